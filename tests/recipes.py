"""Seeded input recipes shared by tests/golden/make_golden.py and the parity tests.

Each recipe mirrors how the reference harness builds its inputs (cited per function) but is
written for this repo; torch's CPU generator is deterministic for a fixed torch build, and every
golden file also stores a sha256 of the inputs it was generated from so drift is detected rather
than silently compared.
"""
import math

import torch

FP8_MAX, FP8_MIN = 400.0, -400.0          # reference clamps to +-400 before the e4m3 cast
SCALE_FACTOR = 1e-3                        # block-scale magnitude used by the reference tests

# name, M, N, K, E, topk, block_n, block_k, masked(-1 ids), seed, store_full_inputs
MOE_FP8_CASES = [
    # shapes of /root/reference/test_moe_fp8_ext.py:122-124 (block [64,128])
    ("m2_n128_k128_e8_t4", 2, 128, 128, 8, 4, 64, 128, False, 1111, True),
    ("m121_n512_k1024_e8_t2", 121, 512, 1024, 8, 2, 64, 128, False, 1112, False),
    ("m1212_n512_k1024_e8_t2", 1212, 512, 1024, 8, 2, 64, 128, False, 1113, False),
    # shape of /root/reference/test_moe_offloading_cpu.py:145 (block [128,128], -1 padded ids)
    ("masked_m14_n128_k128_e8_t8", 14, 128, 128, 8, 8, 128, 128, True, 1114, True),
    ("masked_m300_n256_k512_e16_t8", 300, 256, 512, 16, 8, 128, 128, True, 1115, False),
    # Qwen3-30B-A3B expert dims (models/Qwen3-VL-30B-A3B-Instruct/config.json:16,22,25,26), few experts
    ("qwen3dims_m96_e8_t8", 96, 768, 2048, 8, 8, 128, 128, False, 1116, False),
    # ragged / edge: one token, topk == E, every expert hit
    ("m1_n128_k256_e4_t4", 1, 128, 256, 4, 4, 128, 128, False, 1117, True),
]

# name, M, N, K, E, topk, seed, full      (/root/reference/test_moe_int8.py:139-141)
MOE_INT8_CASES = [
    ("m1_n128_k256_e8_t2", 1, 128, 256, 8, 2, 2111, True),
    ("m39_n1280_k1024_e8_t3", 39, 1280, 1024, 8, 3, 2112, False),
    ("m257_n256_k512_e8_t3", 257, 256, 512, 8, 3, 2113, False),
]

# name, M, N, K, E, topk, renormalize, seed, full   (/root/reference/test_moe.py:109-112)
MOE_BF16_CASES = [
    ("m4_n32_k32_e4_t2", 4, 32, 32, 4, 2, False, 3111, True),
    ("m2_n128_k32_e4_t2_renorm", 2, 128, 32, 4, 2, True, 3112, True),
    ("m224_n256_k1056_e8_t2_renorm", 224, 256, 1024 + 32, 8, 2, True, 3113, False),
]

# name, M, E, G, topk, topk_group, renormalize, biased, seed
# (/root/reference/test_grouped_topk.py:80-88, test_biased_grouped_topk.py:94-95)
TOPK_CASES = [
    ("m123_e8_g2_k2_tg1_r1", 123, 8, 2, 2, 1, True, False, 111),
    ("m123_e16_g4_k3_tg2_r0", 123, 16, 4, 3, 2, False, False, 112),
    ("m123_e32_g4_k3_tg2_r1", 123, 32, 4, 3, 2, True, False, 113),
    ("m123_e64_g1_k6_tg1_r0", 123, 64, 1, 6, 1, False, False, 114),
    ("m123_e256_g8_k4_tg8_r1", 123, 256, 8, 4, 8, True, False, 115),
    ("m123_e160_g8_k6_tg2_r0", 123, 160, 8, 6, 2, False, False, 116),
    ("m123_e128_g1_k8_tg1_r1", 123, 128, 1, 8, 1, True, False, 117),   # Qwen3 routing: 128 experts top-8
    ("biased_m122_e256_g8_k8_tg2_r1", 122, 256, 8, 8, 2, True, True, 118),
    ("biased_m122_e256_g8_k8_tg2_r0", 122, 256, 8, 8, 2, False, True, 119),
]


def _gen(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return g


def routing_softmax_topk(M, E, topk, g, dtype=torch.bfloat16):
    """softmax(randn[M,E] in bf16, f32) -> torch.topk; /root/reference/test_moe_fp8_ext.py:110-112."""
    score = torch.randn(M, E, generator=g).to(dtype)
    score = torch.softmax(score, dim=-1, dtype=torch.float32)
    w, ids = torch.topk(score, topk)
    return w.contiguous(), ids.to(torch.int32).contiguous()


def fp8_weight(shape, g):
    """randn*400 clamped then cast to e4m3fn; /root/reference/test_moe_fp8_ext.py:98-102."""
    return (torch.randn(*shape, generator=g) * FP8_MAX).clamp(min=FP8_MIN, max=FP8_MAX).to(torch.float8_e4m3fn)


def moe_fp8_inputs(M, N, K, E, topk, bn, bk, masked, seed):
    g = _gen(seed)
    if masked:
        # /root/reference/test_moe_offloading_cpu.py:57-74: random ids, keep with prob ratio, pad -1,
        # drop tokens whose slots are all -1, random (signed) routing weights
        ids = torch.randint(0, E, (M, topk), generator=g, dtype=torch.int32)
        ratio = max(topk / 128.0, 0.25)
        keep = torch.rand(M, topk, generator=g) < ratio
        ids[~keep] = -1
        rows = ~(ids == -1).all(dim=1)
        ids = ids[rows].contiguous()
        M = ids.shape[0]
        assert M > 0
        topk_weight = torch.randn(M, topk, generator=g)
    a = (torch.randn(M, K, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    w1 = fp8_weight((E, 2 * N, K), g)
    w2 = fp8_weight((E, K, N), g)
    w1s = torch.randn(E, 2 * N // bn, K // bk, generator=g) * SCALE_FACTOR
    w2s = torch.randn(E, K // bn, N // bk, generator=g) * SCALE_FACTOR
    if not masked:
        topk_weight, ids = routing_softmax_topk(M, E, topk, g)
    return dict(a=a, w1=w1, w2=w2, w1s=w1s, w2s=w2s, topk_weight=topk_weight, topk_ids=ids)


def moe_int8_inputs(M, N, K, E, topk, seed):
    """/root/reference/test_moe_int8.py:97-124."""
    g = _gen(seed)
    a = (torch.randn(M, K, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    w1 = (((torch.rand(E, 2 * N, K, generator=g) - 0.5) * 2) * 127).clamp(-128, 127).to(torch.int8)
    w2 = (((torch.rand(E, K, N, generator=g) - 0.5) * 2) * 127).clamp(-128, 127).to(torch.int8)
    w1s = torch.rand(E, 2 * N, generator=g) * 1e-2
    w2s = torch.rand(E, K, generator=g) * 1e-2
    topk_weight, ids = routing_softmax_topk(M, E, topk, g)
    return dict(a=a, w1=w1, w2=w2, w1s=w1s, w2s=w2s, topk_weight=topk_weight, topk_ids=ids)


def moe_bf16_inputs(M, N, K, E, topk, seed):
    """/root/reference/test_moe.py:96-101."""
    g = _gen(seed)
    a = (torch.randn(M, K, generator=g) / 10).to(torch.bfloat16)
    w1 = (torch.randn(E, 2 * N, K, generator=g) / 10).to(torch.bfloat16)
    w2 = (torch.randn(E, K, N, generator=g) / 10).to(torch.bfloat16)
    score = torch.randn(M, E, generator=g).to(torch.bfloat16)
    return dict(a=a, w1=w1, w2=w2, score=score)


def topk_inputs(M, E, biased, seed):
    """/root/reference/test_grouped_topk.py:44-46, test_biased_grouped_topk.py:53-55."""
    g = _gen(seed)
    hidden = torch.randn(M, 100, generator=g).to(torch.bfloat16)
    gating = (torch.randn(M, E, generator=g) * 2 * M).to(torch.bfloat16)
    d = dict(hidden=hidden, gating=gating)
    if biased:
        d["bias"] = torch.randn(E, generator=g).to(torch.bfloat16)
    return d


# ---- dense GEMMs ---------------------------------------------------------------------------------------------------
# name, M, N, K, has_bias, seed           int8 w8a8 per-token / per-channel (/root/reference/test_gemm_int8.py:75-78)
GEMM_INT8_CASES = [
    ("m128_n384_k544_bias", 128, 384, 544, True, 4111),
    ("m2_n32_k32", 2, 32, 32, False, 4112),
    ("m300_n1536_k2048", 300, 1536, 2048, False, 4113),     # Qwen3 expert gate_up as a dense shape
]
# name, M, N, K, has_bias, chunk(row-strided mat1), seed     fp8 block [64,128] (/root/reference/test_gemm_fp8.py:109-122)
GEMM_FP8_CASES = [
    ("m1_n128_k512_bias", 1, 128, 512, True, False, 4211),
    ("m111_n192_k640", 111, 192, 640, False, False, 4212),
    ("m14_n192_k768_bias_chunk", 14, 192, 768, True, True, 4213),
    ("m64_n2816_k1024_bias", 64, 2816, 1024, True, False, 4214),
    ("m500_n768_k1024_bias", 500, 768, 1024, True, False, 4215),      # large enough for the tuned 256-token kernel
    ("m333_n512_k2048_chunk", 333, 512, 2048, False, True, 4216),
]
# name, M, N, K, has_bias, seed           bf16 weight_packed_linear (/root/reference/test_gemm.py:30-33)
GEMM_BF16_CASES = [
    ("m1_n416_k512_bias", 1, 416, 512, True, 4311),
    ("m11_n416_k512_bias", 11, 416, 512, True, 4312),
    ("m128_n4096_k4096", 128, 4096, 4096, False, 4313),     # BASELINE.json config 0
]
# name, m, n, k, rsf, seed                shared expert bf16 + int8 (/root/reference/test_shared_experts.py:85-87)
SHARED_CASES = [
    ("m2_n32_k32", 2, 32, 32, 16.0, 5111),
    ("m121_n128_k64", 121, 128, 64, 16.0, 5112),
]
# name, M, N, K, rsf, seed                shared expert fp8 block [64,128] (/root/reference/test_moe_fp8_ext.py:65-67)
SHARED_FP8_CASES = [
    ("m2_n256_k1024", 2, 256, 1024, 16.0, 5211),
    ("m12_n128_k256", 12, 128, 256, 16.0, 5212),
    ("m1212_n128_k256", 1212, 128, 256, 16.0, 5213),
]


def gemm_int8_inputs(M, N, K, has_bias, seed):
    g = _gen(seed)
    A = (torch.randn(M, K, generator=g) / 10).to(torch.bfloat16)
    B = (torch.rand(N, K, generator=g) - 0.5) * 2
    Bq = (B * 127).clamp(-128, 127).to(torch.int8)
    Bs = torch.rand(N, generator=g) * 1e-2
    d = dict(A=A, Bq=Bq, Bs=Bs)
    if has_bias:
        d["bias"] = torch.randn(N, generator=g)
    return d


def gemm_fp8_inputs(M, N, K, has_bias, chunk, seed, bn=64, bk=128):
    g = _gen(seed)
    if chunk:
        data = torch.randn(M, K + 6, generator=g).to(torch.bfloat16).narrow(1, 0, K)
    else:
        data = torch.randn(M, K, generator=g).to(torch.bfloat16)
    data = data / math.sqrt(K)
    w = fp8_weight((N, K), g)
    scales = torch.randn(N // bn, K // bk, generator=g) * SCALE_FACTOR
    d = dict(data=data, w=w, scales=scales)
    if has_bias:
        d["bias"] = torch.randn(N, generator=g)
    return d


def gemm_bf16_inputs(M, N, K, has_bias, seed):
    g = _gen(seed)
    d = dict(mat1=torch.randn(M, K, generator=g).to(torch.bfloat16), mat2=torch.randn(N, K, generator=g).to(torch.bfloat16))
    if has_bias:
        d["bias"] = torch.randn(N, generator=g)
    return d


def shared_inputs(m, n, k, seed):
    """/root/reference/test_shared_experts.py:57-60."""
    g = _gen(seed)
    return dict(hs=(torch.randn(m, k, generator=g) / k).to(torch.bfloat16),
                w1=torch.randn(2 * n, k, generator=g).to(torch.bfloat16),
                w2=torch.randn(k, n, generator=g).to(torch.bfloat16),
                fused=(torch.randn(m, k, generator=g) / k).to(torch.bfloat16))


def shared_fp8_inputs(M, N, K, seed, bn=64, bk=128):
    """/root/reference/test_moe_fp8_ext.py:27-49."""
    g = _gen(seed)
    return dict(a=(torch.randn(M, K, generator=g) / math.sqrt(K)).to(torch.bfloat16),
                w1=fp8_weight((2 * N, K), g), w2=fp8_weight((K, N), g),
                w1s=torch.randn(2 * N // bn, K // bk, generator=g) * SCALE_FACTOR,
                w2s=torch.randn(K // bn, N // bk, generator=g) * SCALE_FACTOR,
                fused=(torch.randn(M, K, generator=g) / math.sqrt(K)).to(torch.bfloat16))


# ---- row kernels -----------------------------------------------------------------------------------------------------
# name, rows, hidden, dtype, seed          (/root/reference/test_norm.py:64-65)
NORM_CASES = [
    ("r64_h4096_bf16", 64, 4096, torch.bfloat16, 6111),
    ("r33_h4109_f16", 33, 4096 + 13, torch.float16, 6112),
    ("r5_h5120_f16", 5, 5120, torch.float16, 6113),
]
# name, rows, two_d, dtype, seed           (/root/reference/test_activation.py:30-31, bench_silu_and_mul.py:60-65)
ACT_CASES = [
    ("r16_d22016_bf16", 16, 22016, torch.bfloat16, 6211),
    ("r17_d22016_f16", 17, 22016, torch.float16, 6212),
    ("r3_d36864_bf16", 3, 36864, torch.bfloat16, 6213),
    ("r7_d74_f16", 7, 74, torch.float16, 6214),
]


def norm_inputs(rows, hidden, dtype, seed):
    g = _gen(seed)
    return dict(x=torch.randn(rows, hidden, generator=g).to(dtype), w=torch.randn(hidden, generator=g).to(dtype),
                res=torch.randn(rows, hidden, generator=g).to(dtype))


def act_inputs(rows, two_d, dtype, seed):
    g = _gen(seed)
    return dict(x=torch.randn(rows, two_d, generator=g).to(dtype))


# ---- attention ---------------------------------------------------------------------------------------------------------
# name, B, N_CTX, H_Q, H_KV, D, DV, mla(buffer has 1 head, no prefix), seed     (/root/reference/test_extend.py:190-198)
EXTEND_CASES = [
    ("b1_ctx23_hq8_hkv2", 1, 23, 8, 2, 128, 96, False, 7111),
    ("b1_ctx123_hq16_hkv1_mla", 1, 123, 16, 1, 128, 96, True, 7112),
    ("b4_ctx1230_hq16_hkv4", 4, 1230, 16, 4, 128, 96, False, 7113),
    ("b4_ctx1230_hq16_hkv16_mla", 4, 1230, 16, 16, 128, 96, True, 7114),
    ("b2_ctx700_hq32_hkv4_d128", 2, 700, 32, 4, 128, 128, False, 7115),       # Qwen3-30B-A3B attention heads
    ("b2_ctx300_hq22_hkv22_d192", 2, 300, 22, 22, 192, 128, True, 7116),      # bench_extend.py:111-112 MLA prefill dims
]
# name, B, H_Q, H_KV, D, DV, seq_len, v_alias, seed      (/root/reference/test_mla.py:178-183, test_decoding.py:151-167)
DECODE_CASES = [
    ("mla_b1_s888", 1, 22, 1, 576, 512, 8 * 111, True, 7211),
    ("mla_b4_s1024", 4, 22, 1, 576, 512, 8 * 128, True, 7212),
    ("mla_b40_s1064", 40, 22, 1, 576, 512, 8 * 133, True, 7213),
    ("gqa_b1_hq40_hkv8_s1024", 1, 40, 8, 128, 128, 1024, False, 7214),
    ("gqa_b3_hq32_hkv4_s333", 3, 32, 4, 128, 128, 333, False, 7215),
]


# The sizes BASELINE.json config 3 names (seqlen <= 8k), as /root/reference/bench_extend.py:107-112 and
# /root/reference/test_mla.py:178-186 run them: prefix and extend fifty-fifty (MLA: no prefix), fixed lengths.  Too large to
# store: the goldens keep the reference oracle's output for a sample of token rows (`rows`), the inputs come from the seed.
# name, B, N_CTX, H_Q, H_KV, D, DV, mla, seed
EXTEND_BIG_CASES = [
    ("bench_b1_ctx4096_hq32_hkv4", 1, 4096, 32, 4, 128, 128, False, 7121),     # Qwen3-30B-A3B heads, bench_extend.py:107
    ("bench_b1_ctx8192_hq16_hkv2", 1, 8192, 16, 2, 128, 128, False, 7122),     # bench_extend.py:108
    ("bench_b4_ctx3500_hq22_mla_d192", 4, 3500, 22, 22, 192, 128, True, 7123),  # bench_extend.py:111-112
]
# name, B, H_Q, H_KV, D, DV, seq_len, v_alias, seed
DECODE_BIG_CASES = [
    ("mla_b40_s4096", 40, 22, 1, 576, 512, 4096, True, 7221),
    ("gqa_b64_hq32_hkv4_s4096", 64, 32, 4, 128, 128, 4096, False, 7222),
]


def extend_inputs_fixed(B, N_CTX, H_Q, H_KV, D, DV, mla, seed):
    """/root/reference/bench_extend.py:22-56: every sequence has prefix N_CTX/2 + extend N_CTX/2 (MLA: extend N_CTX)."""
    g = _gen(seed)
    dt = torch.bfloat16
    prefix = torch.full((B,), 0 if mla else N_CTX // 2, dtype=torch.int32)
    extend = torch.full((B,), N_CTX if mla else N_CTX // 2, dtype=torch.int32)
    seq = prefix + extend
    req_to_tokens = torch.zeros(B, int(seq.max()), dtype=torch.int32)
    start = torch.zeros(B, dtype=torch.int32)
    start[1:] = torch.cumsum(seq[:-1], 0)
    start_ext = torch.zeros(B, dtype=torch.int32)
    start_ext[1:] = torch.cumsum(extend[:-1], 0)
    for i in range(B):
        req_to_tokens[i, :seq[i]] = torch.arange(int(start[i]), int(start[i] + seq[i]))
    total, ext_total = int(seq.sum()), int(extend.sum())
    HB = 1 if mla else H_KV
    k_buffer = torch.randn(total, HB, D, generator=g).to(dt)
    v_buffer = torch.randn(total, HB, DV, generator=g).to(dt)
    k_extend = torch.empty(ext_total, H_KV, D, dtype=dt)
    v_extend = torch.empty(ext_total, H_KV, DV, dtype=dt)
    q_extend = torch.randn(ext_total, H_Q, D, generator=g).to(dt)
    for i in range(B):
        s0, e0 = int(start[i] + prefix[i]), int(start[i] + seq[i])
        s1, e1 = int(start_ext[i]), int(start_ext[i] + extend[i])
        k_extend[s1:e1] = k_buffer[s0:e0]
        v_extend[s1:e1] = v_buffer[s0:e0]
    return dict(q_extend=q_extend, k_extend=k_extend, v_extend=v_extend, k_buffer=k_buffer, v_buffer=v_buffer,
                req_to_tokens=req_to_tokens, b_req_idx=torch.arange(B, dtype=torch.int64), b_seq_len=seq.to(torch.int64),
                b_prefix=prefix, b_extend=extend, b_start_loc_extend=start_ext)


def sample_rows(n, count, seed):
    """`count` row indices of 0..n-1: both ends, both sides of every 128-row block edge near the ends, random in between."""
    g = _gen(seed)
    fixed = [0, 1, 127, 128, 129, n // 2 - 1, n // 2, n - 130, n - 129, n - 128, n - 2, n - 1]
    fixed = sorted({r for r in fixed if 0 <= r < n})
    rest = torch.randperm(n, generator=g)[: max(0, count - len(fixed))].tolist()
    return torch.tensor(sorted(set(fixed) | set(rest)), dtype=torch.long)


def extend_inputs(B, N_CTX, H_Q, H_KV, D, DV, mla, seed):
    """/root/reference/test_extend.py:79-138."""
    g = _gen(seed)
    dt = torch.bfloat16
    prefix = torch.randint(1, max(N_CTX // 2, 2), (B,), generator=g, dtype=torch.int32)
    if mla:
        prefix.zero_()
    extend = torch.randint(1, max(N_CTX // 2, 2), (B,), generator=g, dtype=torch.int32)
    seq = prefix + extend
    req_to_tokens = torch.zeros(B, int(seq.max()), dtype=torch.int32)
    start = torch.zeros(B, dtype=torch.int32)
    start[1:] = torch.cumsum(seq[:-1], 0)
    start_ext = torch.zeros(B, dtype=torch.int32)
    start_ext[1:] = torch.cumsum(extend[:-1], 0)
    for i in range(B):
        req_to_tokens[i, :seq[i]] = torch.arange(int(start[i]), int(start[i] + seq[i]))
    total, ext_total = int(seq.sum()), int(extend.sum())
    HB = 1 if mla else H_KV
    k_buffer = torch.randn(total, HB, D, generator=g).to(dt)
    v_buffer = torch.randn(total, HB, DV, generator=g).to(dt)
    k_extend = torch.empty(ext_total, H_KV, D, dtype=dt)
    v_extend = torch.empty(ext_total, H_KV, DV, dtype=dt)
    q_extend = torch.randn(ext_total, H_Q, D, generator=g).to(dt)
    for i in range(B):
        s0, e0 = int(start[i] + prefix[i]), int(start[i] + seq[i])
        s1, e1 = int(start_ext[i]), int(start_ext[i] + extend[i])
        k_extend[s1:e1] = k_buffer[s0:e0]
        v_extend[s1:e1] = v_buffer[s0:e0]
    return dict(q_extend=q_extend, k_extend=k_extend, v_extend=v_extend, k_buffer=k_buffer, v_buffer=v_buffer,
                req_to_tokens=req_to_tokens, b_req_idx=torch.arange(B, dtype=torch.int64), b_seq_len=seq.to(torch.int64),
                b_prefix=prefix, b_extend=extend, b_start_loc_extend=start_ext)


def decode_inputs(B, H_Q, H_KV, D, DV, seq_len, v_alias, seed):
    """/root/reference/test_mla.py:68-105."""
    g = _gen(seed)
    dt = torch.bfloat16
    total = B * seq_len
    q = torch.randn(B, H_Q, D, generator=g).to(dt)
    k_buffer = torch.randn(total, H_KV, D, generator=g).to(dt)
    key = torch.randn(B, H_KV, D, generator=g).to(dt)
    if v_alias:
        v_buffer, value = None, None     # views of k_buffer / key are made by the caller
    else:
        v_buffer = torch.randn(total, H_KV, DV, generator=g).to(dt)
        value = torch.randn(B, H_KV, DV, generator=g).to(dt)
    loc = torch.randperm(total, generator=g)[:B].to(torch.int64)
    req_to_token = torch.arange(total).reshape(B, seq_len).to(torch.int64)
    return dict(q=q, k_buffer=k_buffer, v_buffer=v_buffer, key=key, value=value, loc=loc, req_to_token=req_to_token,
                b_req_idx=torch.arange(B, dtype=torch.int64), b_seq_len=torch.full((B,), seq_len, dtype=torch.int64))


# name, B, hidden, seed     qkv_proj_with_rope (/root/reference/test_absorb.py:111-131,196; DeepSeek-style MLA dims :11-17)
ABSORB_DIMS = dict(kv_lora_rank=512, qk_nope_head_dim=128, qk_rope_head_dim=64, num_heads=22, q_lora_rank=1536)
ABSORB_CASES = [
    ("b3_h7168", 3, 7168, 9111),
    ("b1_h2048", 1, 2048, 9112),
    ("b17_h1024", 17, 1024, 9113),
]


def absorb_inputs(B, hidden, seed):
    """/root/reference/test_absorb.py:113-131.  int8 variants of the three projections are quantised per output row
    like the reference does (:163-165)."""
    d = ABSORB_DIMS
    g = _gen(seed)
    bf = torch.bfloat16
    H, qk_head = d["num_heads"], d["qk_nope_head_dim"] + d["qk_rope_head_dim"]
    out = dict(
        hidden_states=(torch.randn(B, hidden, generator=g) / hidden).to(bf),
        q_a_proj_weight=(torch.randn(d["q_lora_rank"], hidden, generator=g) * 0.1).to(bf),
        norm_weight1=torch.randn(d["q_lora_rank"], generator=g).to(bf),
        q_b_proj_weight=(torch.randn(H * qk_head, d["q_lora_rank"], generator=g) * 0.1).to(bf),
        w_kc=(torch.randn(H, d["kv_lora_rank"], d["qk_nope_head_dim"], generator=g) * 0.1).to(bf),
        kv_a_proj_weight=(torch.randn(d["kv_lora_rank"] + d["qk_rope_head_dim"], hidden, generator=g) * 0.1).to(bf),
        norm_weight2=torch.randn(d["kv_lora_rank"], generator=g).to(bf),
        pos=torch.randint(10, 100, (B,), generator=g),
        cos_sin_cache=torch.randn(100, d["qk_rope_head_dim"], generator=g).to(bf),
    )
    return out


# name, batch, max_q, max_k, H, Hkv, D, DV, causal, varlen, seed     (/root/reference/test_flash_attn_varlen.py:111-116, bench :160)
VARLEN_CASES = [
    ("b1_q123_k45_h1_d128_dv96", 1, 123, 45, 1, 1, 128, 96, False, False, 9211),
    ("b4_h32x4_d64_dv94", 4, 160, 60, 32, 4, 64, 94, False, True, 9212),
    ("b4_h32x4_d64_dv72_causal", 4, 160, 60, 32, 4, 64, 72, True, True, 9213),
    ("b4_h32x4_d64_dv80_causal", 4, 160, 60, 32, 4, 64, 80, True, True, 9214),
    ("b4_h32x4_d64_dv96_causal_fixed", 4, 160, 60, 32, 4, 64, 96, True, False, 9215),
    ("b4_h32x4_d64_dv96_fixed", 4, 160, 60, 32, 4, 64, 96, False, False, 9216),
    ("b3_q700_k700_h6_d72_causal", 3, 700, 700, 6, 6, 72, 72, True, True, 9217),
]


def varlen_inputs(batch, max_q, max_k, H, Hkv, D, DV, varlen, seed):
    """/root/reference/test_flash_attn_varlen.py:63-86."""
    g = _gen(seed)
    if varlen:
        sq = torch.randint(1, max_q, (batch,), generator=g, dtype=torch.int32)
        sk = torch.randint(1, max_k, (batch,), generator=g, dtype=torch.int32)
    else:
        sq = torch.full((batch,), max_q, dtype=torch.int32)
        sk = torch.full((batch,), max_k, dtype=torch.int32)
    cu_q = torch.zeros(batch + 1, dtype=torch.int32)
    cu_k = torch.zeros(batch + 1, dtype=torch.int32)
    cu_q[1:] = torch.cumsum(sq, 0)
    cu_k[1:] = torch.cumsum(sk, 0)
    bf = torch.bfloat16
    return dict(q=torch.randn(int(sq.sum()), H, D, generator=g).to(bf), k=torch.randn(int(sk.sum()), Hkv, D, generator=g).to(bf),
                v=torch.randn(int(sk.sum()), Hkv, DV, generator=g).to(bf), cu_q=cu_q, cu_k=cu_k, max_q=int(sq.max()),
                max_k=int(sk.max()))


# name, B, M, N, K, chunk, seed      (/root/reference/test_bmm_fp8.py:122-126: m in {1, 2, 11, 111}, two shapes, + (1, 5, 96, 160))
BMM_CASES = [(f"b17_m{m}_n160_k544", 17, m, 160, 544, True, 9400 + m) for m in (1, 2, 11, 111)] + \
            [(f"b16_m{m}_n512_k160", 16, m, 512, 160, True, 9500 + m) for m in (1, 2, 11, 111)] + \
            [("b1_m5_n96_k160", 1, 5, 96, 160, True, 9600), ("b3_m7_n40_k72_plain", 3, 7, 40, 72, False, 9601)]


def bmm_inputs(B, M, N, K, chunk, seed):
    """/root/reference/test_bmm_fp8.py:43-50: mat1 [B, M, K] as a transposed (and, with `chunk`, narrowed) view of
    [M, B, K(+64)]; mat2 [B, N, K] contiguous; out [B, M, N] the same kind of view."""
    g = _gen(seed)
    bf = torch.bfloat16
    pad = 64 if chunk else 0
    mat1 = torch.randn(M, B, K + pad, generator=g).to(bf).narrow(2, 0, K).transpose(0, 1)
    mat2 = torch.randn(B, N, K, generator=g).to(bf)
    out = torch.zeros(M, B, N + pad, dtype=bf).narrow(2, 0, N).transpose(0, 1)
    return dict(mat1=mat1, mat2=mat2, out=out)


# name, M, N, K, kind, bias, seed    (/root/reference/test_mxfp4.py:206-210; "quant" = weights quantised from floats (:150-157),
# "raw" = random nibbles with scale bytes 126 (:178-181); the last two widen the scale range and the row count)
MXFP4_CASES = [
    ("m1_n32_k32_quant", 1, 32, 32, "quant", False, 9701),
    ("m1_n32_k2048_quant", 1, 32, 2048, "quant", False, 9702),
    ("m112_n960_k352_raw", 112, 960, 352, "raw", False, 9703),
    ("m2_n128_k128_raw_bias", 2, 128, 128, "raw", True, 9704),
    ("m11_n64_k96_raw", 11, 64, 96, "raw", False, 9705),
    ("m300_n256_k512_wide_bias", 300, 256, 512, "wide", True, 9706),
]


def mxfp4_inputs(M, N, K, kind, has_bias, seed, quantize=None):
    """/root/reference/test_mxfp4.py:146-164,174-194.  `quantize` = the MX-fp4 quantiser for kind "quant" (the golden
    generator passes the reference's, tests pass the oracle's restatement; the golden file also stores wq / ws)."""
    g = _gen(seed)
    a = (torch.randn(M, K, generator=g) / 10).to(torch.bfloat16)
    if kind == "quant":
        b = ((torch.rand(N, K, generator=g) - 0.5) * 2).to(torch.bfloat16) * 1e-2
        wq, ws = quantize(b)
    else:
        wq = torch.randint(0, 256, (N, K // 2), generator=g, dtype=torch.uint8)
        lo, hi = (126, 127) if kind == "raw" else (110, 131)
        ws = torch.randint(lo, hi, (N, K // 32), generator=g, dtype=torch.uint8)
    bias = torch.randn(N, generator=g) if has_bias else None
    return dict(a=a, wq=wq, ws=ws.view(N, K // 32), bias=bias)
