"""W8A16 fused_experts on the block-scaled fp8 matrix cores with the bf16 activations as two exact e4m3 terms
(sgl-cpu-tests_amd/csrc/moe_gemm_fp8w_s128.hip, fp8_split.h; the default from 64 rows per expert on).

Same operator, same oracle and the same pass criteria as the bf16-MFMA kernel (/root/reference/test_moe_fp8_ext.py:70-91,
118-120; utils.compare): the split changes how the products are formed, not what is computed -- x == hi + lo exactly for every
element within 2^13 of its 128-block's largest magnitude."""
import ctypes

import pytest
import torch

import recipes
from conftest import load_golden
from oracle import c_oracle, moe_a8

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import sgl_kernel  # noqa: F401
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.ops.sgl_kernel


def e4m3_value(b):
    return b.view(torch.float8_e4m3fn).float()


def gpu_split(x):
    from sgl_kernel import _lib
    rows, cols = x.shape
    q = torch.empty(rows, 2 * cols, dtype=torch.uint8, device="cuda")
    ss = (cols // 128 + 3) // 4 * 4
    s = torch.zeros(rows, ss, dtype=torch.uint8, device="cuda")
    rc = _lib.lib().sglk_split_fp8_block128(ctypes.c_void_p(x.data_ptr()), x.stride(0), ctypes.c_void_p(q.data_ptr()), 2 * cols,
                                            ctypes.c_void_p(s.data_ptr()), ss, rows, cols,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "split_fp8_block128")
    torch.cuda.synchronize()
    return q.cpu(), s.cpu()[:, :cols // 128]


def reconstruct(q, s, cols):
    """hi * 2^(sb-127) + lo * 2^(sb-131), columns put back into natural k order."""
    rows = q.shape[0]
    g = q.view(rows, cols // 64, 2, 64)
    order = torch.from_numpy(moe_a8.packed_k_order(64))
    inv = torch.empty(64, dtype=torch.long)
    inv[order] = torch.arange(64)                     # natural k -> storage position
    hi = e4m3_value(g[:, :, 0, :])[:, :, inv].reshape(rows, cols)
    lo = e4m3_value(g[:, :, 1, :])[:, :, inv].reshape(rows, cols)
    sc = torch.ldexp(torch.ones(rows, cols // 128), s.int() - 127).repeat_interleave(128, dim=1)
    return hi * sc + lo * sc / 16, hi * sc


@pytest.mark.parametrize("rows,cols", [(3, 128), (41, 2048), (7, 768), (2, 4096 + 256)])
def test_split_is_exact(ops, rows, cols):
    """x == hi + lo bit for bit for every element within 2^13 of its block's largest magnitude (hi is then a normal e4m3 number),
    incl. negative values, zeros and block maxima on the scale boundary; the error bound 2^-17 * amax(block) below that, on a
    block spanning 2^30."""
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols, generator=g) * torch.exp2(torch.randint(-18, 10, (rows, 1), generator=g).float())
    x[0, :64] = 0.0
    x[1, 5], x[1, 6] = 1.75, -3.5
    xb = x.bfloat16()
    q, s = gpu_split(xb.cuda())
    rec, hi = reconstruct(q, s, cols)
    blk_amax = xb.float().abs().reshape(rows, cols // 128, 128).amax(-1)
    in_range = xb.float().abs() >= (blk_amax * 2.0 ** -13).repeat_interleave(128, dim=1)
    assert torch.equal(rec[in_range], xb.float()[in_range]), "hi + lo != x inside the exact range"
    assert (hi.abs() <= 448 * torch.ldexp(torch.ones(rows, cols // 128), s.int() - 127).repeat_interleave(128, 1)).all()
    # a block with a 2^30 spread: everything below 2^-14 of the maximum may lose bits, never more than the stated bound
    y = torch.randn(2, 256, generator=g).bfloat16()
    y[:, 0] = 2.0 ** 20
    y[:, 1:] = (y[:, 1:].float() * torch.exp2(-torch.arange(255).float() / 8.0)).bfloat16()
    q, s = gpu_split(y.cuda())
    rec, _ = reconstruct(q, s, 256)
    amax = y.float().abs().reshape(2, 2, 128).amax(-1).repeat_interleave(128, dim=1)
    assert ((rec - y.float()).abs() <= amax * 2.0 ** -17).all()


@pytest.fixture(scope="module")
def qwen3(ops):
    import test_moe_fp8_bench_path_gpu as bp
    return bp.make_qwen3(ops)


def run(ops, inp, block):
    d = {k: v.cuda() for k, v in inp.items()}
    w1p, w2p = ops.convert_weight_packed(d["w1"]), ops.convert_weight_packed(d["w2"])
    out = ops.fused_experts_cpu(d["a"], w1p, w2p, d["topk_weight"], d["topk_ids"], False, False, True,
                                d["w1s"], d["w2s"], list(block), None, None, True)
    torch.cuda.synchronize()
    return out


def check_close(out_bf16, ref_f32, what):
    out = out_bf16.float().cpu()
    assert torch.allclose(ref_f32.bfloat16(), out_bf16.cpu(), rtol=1e-2, atol=1e-2), f"{what}: reference predicate failed"
    err = (out - ref_f32).norm() / ref_f32.norm().clamp_min(1e-12)
    assert err < 6e-3, f"{what}: relative RMS error {err:.2e}"
    return float(err)


# ---- the split on 128-token tiles, four waves, two workgroups per CU (sgl-cpu-tests_amd/csrc/moe_gemm_fp8w_s128.hip) ----------

def s128_taken():
    from sgl_kernel import _lib, _ops
    return bool(_ops.last_path & _lib.PATH_SPLIT) and (_ops.last_path & _lib.PATH_TILE_MASK) == 128


@pytest.mark.parametrize("name", ["m1212_n512_k1024_e8_t2", "masked_m300_n256_k512_e16_t8", "qwen3dims_m96_e8_t8"])
def test_s128_kernel_golden(ops, knob, name):
    """Golden cases of the reference's own oracle (tests/golden/make_golden.py), reference predicate + relative RMS, run-to-run
    bit identity, and agreement with the 256-row bf16-MFMA kernel (other rounding points, same stated bound)."""
    case = next(c for c in recipes.MOE_FP8_CASES if c[0] == name)
    _, M, N, K, E, topk, bn, bk, masked, seed, _full = case
    g, _ = load_golden("moe_fp8_" + name)
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, masked, seed)
    knob(SGLK_S128=1, SGLK_MOE_TILE_M=256, SGLK_TAIL_SPLIT=0)
    out = run(ops, inp, (bn, bk))
    assert s128_taken()
    e_s = check_close(out, g["ref_out_f32"], name + " (s128)")
    again = run(ops, inp, (bn, bk))
    assert torch.equal(out, again), "run-to-run bit identity"
    knob(SGLK_S128=0)
    ref_k = run(ops, inp, (bn, bk))
    assert not s128_taken()
    e_bf16 = check_close(ref_k, g["ref_out_f32"], name + " (bf16 MFMA)")
    rel = (out.float() - ref_k.float()).norm() / ref_k.float().norm()
    print(f"[s128] {name}: rel RMS vs oracle {e_s:.2e} (bf16-MFMA kernel {e_bf16:.2e}); between the two kernels {rel:.2e}")
    assert rel < 3e-3 and e_s < e_bf16 * 1.25 + 1e-4


def test_s128_kernel_scale_extremes_and_wide_activations(ops, knob):
    knob(SGLK_S128=1, SGLK_MOE_TILE_M=256, SGLK_TAIL_SPLIT=0)
    M, N, K, E, topk, bn, bk = 1531, 256, 512, 8, 4, 128, 128
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, False, 9001)
    g = torch.Generator().manual_seed(5)
    inp["w1s"] = inp["w1s"].sign() * torch.exp2(torch.rand(inp["w1s"].shape, generator=g) * 16 - 12) * 1e-2
    inp["w1s"][0, 0, 0] = 0.0
    inp["w2s"][1, 0, 0] = 0.0
    inp["w2s"][2, 1, 1] = 2.0 ** -9
    a = inp["a"].float()
    a[::7, 3::128] *= 4096.0
    inp["a"] = a.bfloat16()
    ref = c_oracle.fused_experts_fp8(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], (bn, bk), inp["topk_weight"], inp["topk_ids"])
    k = float(2.0 / ref.abs().max())
    inp["topk_weight"] = inp["topk_weight"] * k
    out = run(ops, inp, (bn, bk))
    assert s128_taken()
    check_close(out, ref * k, "s128: scale extremes + wide activations")


@pytest.mark.parametrize("N,K,E,topk,M", [(256, 256, 4, 2, 700),      # both reductions two K blocks long (the shortest legal)
                                          (384, 768, 8, 4, 900),      # three / six blocks: every (kblocks - 2) % 3 phase ...
                                          (512, 1280, 8, 2, 1300),    # ... four / ten
                                          (640, 1024, 4, 2, 777)])    # five / eight; ragged last tiles throughout
def test_s128_kernel_reduction_lengths(ops, knob, N, K, E, topk, M):
    """The kernel rotates three weight-fragment register sets and is compiled per (kblocks - 2) % 3: every phase, the shortest
    reduction and ragged tiles against the C oracle."""
    knob(SGLK_S128=1, SGLK_MOE_TILE_M=256, SGLK_TAIL_SPLIT=0)
    bn, bk = 128, 128
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, False, 77 + N + K)
    ref = c_oracle.fused_experts_fp8(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], (bn, bk), inp["topk_weight"], inp["topk_ids"])
    k = float(2.0 / ref.abs().max())
    inp["topk_weight"] = inp["topk_weight"] * k
    out = run(ops, inp, (bn, bk))
    assert s128_taken()
    check_close(out, ref * k, f"s128 N={N} K={K}")


@pytest.mark.parametrize("M", [4096, 16384])
def test_s128_kernel_bench_sizes(ops, knob, qwen3, M):
    """Qwen3-30B-A3B dims, all 128 experts, the sizes bench.py runs: >= 256 sampled tokens (every expert, full and tail tiles)
    against the C oracle, in place == out of place, and agreement with the bf16-MFMA kernel."""
    import test_moe_fp8_bench_path_gpu as bp
    a, tw, ids = bp.routed_inputs(M, 200 + M)
    knob(SGLK_S128=1)
    q = qwen3
    out = bp.call(ops, q, a, tw, ids)
    assert s128_taken()
    assert torch.isfinite(out.float()).all()
    toks, fulls, tails, hit = bp.sample_tokens(ids, bp.E, tile=128)
    assert hit == bp.E and len(toks) >= 256
    ref = c_oracle.fused_experts_fp8(a[toks].cpu(), q["w1"], q["w2"], q["w1s"].cpu(), q["w2s"].cpu(), (bp.BN, bp.BK),
                                     tw[toks].cpu(), ids[toks].cpu())
    bp.check_close(out[toks], ref, f"s128 qwen3 M={M}")
    again = bp.call(ops, q, a, tw, ids)
    assert torch.equal(out, again), "run-to-run bit identity"
    knob(SGLK_S128=0)
    out_b = bp.call(ops, q, a, tw, ids)
    rel = (out.float() - out_b.float()).norm() / out_b.float().norm()
    assert rel < 3e-3, f"s128 vs bf16-MFMA kernel: {rel:.2e}"
