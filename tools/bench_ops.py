#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench.py): the M sweep of fp8 fused_experts and the other
operators of SURVEY.md §8 at the shapes the reference benches, each against the roofline that bounds it.

    python tools/bench_ops.py [moe|moe_literal|moe_int8|gemm|attn|absorb|rows|all] > gpurun_out/bench_ops.json

Prints one JSON object per line.  HIP-event timing on the current stream, warm-up, rotating clones where the working
set would otherwise sit in the 256 MiB Infinity Cache.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sgl-cpu-tests_amd"))

import torch  # noqa: E402

import sgl_kernel  # noqa: E402,F401

ops = torch.ops.sgl_kernel
PEAK_BF16, PEAK_HBM = 2500.0, 8000.0   # TFLOP/s dense bf16 MFMA, GB/s (MI355X_MICROARCH.md)


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters   # ms


def graph_ms(fn, reps=200):
    """Device time per call: ten calls captured in one hipGraph, replayed -- no host launch overhead in the number."""
    fn(); fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            fn()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps // 10):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps // 10 * 10)



def emit(**kw):
    print(json.dumps(kw), flush=True)


def bench_moe():
    K, N, E, topk = 2048, 768, 128, 8
    g = torch.Generator(device="cuda").manual_seed(1)
    w1 = (torch.randn(E, 2 * N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn)
    w2 = (torch.randn(E, K, N, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn)
    w1s = torch.randn(E, 2 * N // 128, K // 128, device="cuda", generator=g) * 1e-3
    w2s = torch.randn(E, K // 128, N // 128, device="cuda", generator=g) * 1e-3
    w1p = [ops.convert_weight_packed(w1)]
    w2p = [ops.convert_weight_packed(w2)]
    w1p.append(w1p[0].clone())
    w2p.append(w2p[0].clone())
    del w1, w2
    for M in (1, 4, 16, 64, 256, 512, 1024, 2048, 3929, 4096, 8192, 16384, 32768):
        a = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        score = torch.softmax(torch.randn(M, E, device="cuda", generator=g).bfloat16(), dim=-1, dtype=torch.float32)
        tw, ids = torch.topk(score, topk)
        ids = ids.to(torch.int32)
        touched = int(torch.unique(ids).numel())
        ms = timed(lambda i: ops.fused_experts_cpu(a, w1p[i & 1], w2p[i & 1], tw, ids, False, False, True, w1s, w2s,
                                                   [128, 128], None, None, True), 20 if M >= 4096 else 50)
        flop = M * topk * 6 * N * K
        byts = touched * 3 * N * K + 4 * M * K + 8 * M * topk      # weights of the touched experts + in/out
        t_mfma, t_hbm = flop / (PEAK_BF16 * 1e12), byts / (PEAK_HBM * 1e9)
        emit(op="fused_experts_fp8", M=M, experts_touched=touched, ms=round(ms, 4), tflops=round(flop / ms / 1e9, 2),
             tokens_per_s=round(M / ms * 1e3), algorithmic_gb=round(byts / 1e9, 4), gbps=round(byts / ms / 1e6, 1),
             bound="mfma" if t_mfma > t_hbm else "hbm", roofline_frac=round(max(t_mfma, t_hbm) * 1e3 / ms, 4))
        if M <= 64:
            # decode-size calls are bound by the host enqueue of the four launches; the library never allocates or
            # synchronises, so the call can be captured once and replayed as a hipGraph (what a serving loop does)
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for i in range(2):
                        ops.fused_experts_cpu(a, w1p[0], w2p[0], tw, ids, False, False, True, w1s, w2s, [128, 128], None, None, True)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out_g = ops.fused_experts_cpu(a, w1p[0], w2p[0], tw, ids, False, False, True, w1s, w2s, [128, 128], None, None, True)
                msg = timed(lambda i: graph.replay(), 50)
                emit(op="fused_experts_fp8_hipgraph_replay", M=M, ms=round(msg, 4), tokens_per_s=round(M / msg * 1e3),
                     gbps=round(byts / msg / 1e6, 1), roofline_frac=round(max(t_mfma, t_hbm) * 1e3 / msg, 4))
                del graph, out_g
            except Exception as ex:   # capture support differs between torch builds: report, do not fail the sweep
                emit(op="fused_experts_fp8_hipgraph_replay", M=M, error=str(ex)[:200])


def bench_moe_literal():
    """The reference bench's literal shapes (bench_moe.py:144-145): M in {4, 3929}, N=384, K=7168, E=256, top-8, all
    three weight types, prepacked weights, inplace=True like the reference loop."""
    N, K, E, topk = 384, 7168, 256, 8
    g = torch.Generator(device="cuda").manual_seed(5)
    w1f = torch.randn(E, 2 * N, K, device="cuda", generator=g)
    w2f = torch.randn(E, K, N, device="cuda", generator=g)
    packs = {}
    packs["bf16"] = (ops.convert_weight_packed((w1f * 0.02).bfloat16()), ops.convert_weight_packed((w2f * 0.02).bfloat16()), None, None)
    packs["fp8"] = (ops.convert_weight_packed((w1f * 400).clamp(-400, 400).to(torch.float8_e4m3fn)),
                    ops.convert_weight_packed((w2f * 400).clamp(-400, 400).to(torch.float8_e4m3fn)),
                    torch.rand(E, 2 * N // 128, K // 128, device="cuda", generator=g) * 1e-4,
                    torch.rand(E, K // 128, N // 128, device="cuda", generator=g) * 1e-4)
    packs["int8"] = (ops.convert_weight_packed((w1f * 40).clamp(-127, 127).round().to(torch.int8)),
                     ops.convert_weight_packed((w2f * 40).clamp(-127, 127).round().to(torch.int8)),
                     torch.rand(E, 2 * N, device="cuda", generator=g) * 1e-3, torch.rand(E, K, device="cuda", generator=g) * 1e-3)
    del w1f, w2f
    for M in (4, 3929):
        a0 = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        score = torch.softmax(torch.randn(M, E, device="cuda", generator=g).bfloat16(), dim=-1, dtype=torch.float32)
        tw, ids = torch.topk(score, topk)
        ids = ids.to(torch.int32)
        for kind, (w1, w2, s1, s2) in packs.items():
            a = a0.clone()
            if kind == "bf16":
                fn = lambda i: ops.fused_experts_cpu(a, w1, w2, tw, ids, True, False, False, None, None, None, None, None, True)
            elif kind == "int8":
                fn = lambda i: ops.fused_experts_cpu(a, w1, w2, tw, ids, True, True, False, s1, s2, None, None, None, True)
            else:
                fn = lambda i: ops.fused_experts_cpu(a, w1, w2, tw, ids, True, False, True, s1, s2, [128, 128], None, None, True)
            ms = timed(fn, 20)
            emit(op="fused_experts_reference_shape", weights=kind, M=M, N=N, K=K, E=E, topk=topk, ms=round(ms, 4),
                 tflops=round(M * topk * 6 * N * K / ms / 1e9, 2), tokens_per_s=round(M / ms * 1e3))


def bench_moe_offload():
    """The reference's expert-offloading bench (bench_moe_offloading_cpu.py:15-175): 8 resident experts of 128, ids of the
    non-resident ones padded with -1 at ratio topk / num_experts, all-masked rows dropped, M = 64, N = 256, K = 4096,
    prepacked weights, inplace=True.  Also a batch 64 x larger, where the masked slots are the bulk of the id matrix."""
    N, K, E, topk, total_experts = 256, 4096, 8, 8, 128
    g = torch.Generator(device="cuda").manual_seed(7)
    wb1 = ops.convert_weight_packed((torch.randn(E, 2 * N, K, device="cuda", generator=g) * 0.02).bfloat16())
    wb2 = ops.convert_weight_packed((torch.randn(E, K, N, device="cuda", generator=g) * 0.02).bfloat16())
    wf1 = ops.convert_weight_packed((torch.randn(E, 2 * N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
    wf2 = ops.convert_weight_packed((torch.randn(E, K, N, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
    s1 = torch.rand(E, 2 * N // 128, K // 128, device="cuda", generator=g) * 1e-4
    s2 = torch.rand(E, K // 128, N // 128, device="cuda", generator=g) * 1e-4
    for M in (64, 4096):
        ids = torch.randint(0, E, (M, topk), device="cuda", generator=g, dtype=torch.int32)
        keep = torch.rand(M, topk, device="cuda", generator=g) < topk / total_experts
        ids[~keep] = -1
        ids = ids[(ids >= 0).any(dim=1)]
        m_act = ids.shape[0]
        tw = torch.rand(m_act, topk, device="cuda", generator=g)
        a = (torch.randn(m_act, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        valid = int((ids >= 0).sum())
        for kind, fn in (("bf16", lambda i: ops.fused_experts_cpu(a, wb1, wb2, tw, ids, True, False, False, None, None, None, None, None, True)),
                         ("fp8", lambda i: ops.fused_experts_cpu(a, wf1, wf2, tw, ids, True, False, True, s1, s2, [128, 128], None, None, True))):
            ms = timed(fn, 30)
            emit(op="fused_experts_offloading_shape", weights=kind, M=M, rows_after_masking=m_act, valid_slots=valid, N=N, K=K, E=E,
                 topk=topk, ms=round(ms, 4), tflops=round(valid * 6 * N * K / ms / 1e9, 3))


def bench_moe_int8():
    """int8 W8A8 fused_experts (bench_moe.py:89-106) at the Qwen3-30B-A3B expert shape, prepacked weights."""
    K, N, E, topk = 2048, 768, 128, 8
    g = torch.Generator(device="cuda").manual_seed(6)
    w1 = ops.convert_weight_packed(torch.randint(-127, 128, (E, 2 * N, K), device="cuda", generator=g, dtype=torch.int8))
    w2 = ops.convert_weight_packed(torch.randint(-127, 128, (E, K, N), device="cuda", generator=g, dtype=torch.int8))
    w1s = torch.rand(E, 2 * N, device="cuda", generator=g) * 1e-3
    w2s = torch.rand(E, K, device="cuda", generator=g) * 1e-3
    for M in (1024, 4096, 16384):
        a = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        score = torch.softmax(torch.randn(M, E, device="cuda", generator=g).bfloat16(), dim=-1, dtype=torch.float32)
        tw, ids = torch.topk(score, topk)
        ids = ids.to(torch.int32)
        ms = timed(lambda i: ops.fused_experts_cpu(a, w1, w2, tw, ids, False, True, False, w1s, w2s, None, None, None, True), 20)
        flop = M * topk * 6 * N * K
        emit(op="fused_experts_int8", M=M, ms=round(ms, 4), tops=round(flop / ms / 1e9, 2), tokens_per_s=round(M / ms * 1e3),
             roofline_frac=round(flop / ms / 1e9 / (2 * PEAK_BF16), 4), peak_tops=2 * PEAK_BF16, bound="mfma (int8, 2x bf16)")
    del w1, w2
    # bf16 experts (bench_moe.py:65-82), weights in the reference's VNNI-2 packed order
    b1 = ops.convert_weight_packed((torch.randn(E, 2 * N, K, device="cuda", generator=g) * 0.02).bfloat16())
    b2 = ops.convert_weight_packed((torch.randn(E, K, N, device="cuda", generator=g) * 0.02).bfloat16())
    for M in (1024, 4096, 16384):
        a = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        score = torch.softmax(torch.randn(M, E, device="cuda", generator=g).bfloat16(), dim=-1, dtype=torch.float32)
        tw, ids = torch.topk(score, topk)
        ids = ids.to(torch.int32)
        ms = timed(lambda i: ops.fused_experts_cpu(a, b1, b2, tw, ids, False, False, False, None, None, None, None, None, True), 20)
        flop = M * topk * 6 * N * K
        emit(op="fused_experts_bf16", M=M, ms=round(ms, 4), tflops=round(flop / ms / 1e9, 2), tokens_per_s=round(M / ms * 1e3),
             roofline_frac=round(flop / ms / 1e9 / PEAK_BF16, 4), bound="mfma")


def bench_gemm():
    g = torch.Generator(device="cuda").manual_seed(2)
    # bench_gemm.py:147, BASELINE config 0, Qwen3 expert gate_up, Qwen3 dense FFN up / down
    for (M, N, K) in ((1000, 18432, 2560), (128, 4096, 4096), (4096, 1536, 2048), (4096, 12288, 2048), (4096, 2048, 6144)):
        x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        wb = torch.randn(N, K, device="cuda", generator=g).bfloat16()
        wf = (torch.randn(N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn)
        wi = torch.randint(-127, 127, (N, K), device="cuda", generator=g, dtype=torch.int8)
        sc = torch.randn(N // 128, K // 128, device="cuda", generator=g) * 1e-3
        wfp = ops.convert_weight_packed(wf)          # the reference benches prepacked weights (bench_gemm.py:64-69)
        wbp = ops.convert_weight_packed(wb)
        wip = ops.convert_weight_packed(wi)
        si = torch.rand(N, device="cuda", generator=g) * 1e-2
        xq, xs = ops.per_token_quant_int8_cpu(x)
        flop = 2 * M * N * K
        for name, fn in (
                ("weight_packed_linear_bf16_packed", lambda i: ops.weight_packed_linear(x, wbp, None, True)),
                ("weight_packed_linear_bf16_rowmajor", lambda i: ops.weight_packed_linear(x, wb, None, False)),
                ("fp8_scaled_mm_packed", lambda i: ops.fp8_scaled_mm_cpu(x, wfp, sc, [128, 128], None, torch.bfloat16, True)),
                ("fp8_scaled_mm_rowmajor", lambda i: ops.fp8_scaled_mm_cpu(x, wf, sc, [128, 128], None, torch.bfloat16, False)),
                ("int8_scaled_mm_packed", lambda i: ops.int8_scaled_mm_cpu(xq, wip, xs, si, None, torch.bfloat16, True)),
                ("int8_scaled_mm_with_quant_packed", lambda i: ops.int8_scaled_mm_with_quant(x, wip, si, None, torch.bfloat16, True)),
                ("int8_scaled_mm_with_quant_rowmajor", lambda i: ops.int8_scaled_mm_with_quant(x, wi, si, None, torch.bfloat16, False))):
            ms = timed(fn, 10)
            peak = 2 * PEAK_BF16 if name.startswith("int8_scaled_mm") and "rowmajor" not in name else PEAK_BF16   # int8 MFMA: 2x bf16
            emit(op=name, M=M, N=N, K=K, ms=round(ms, 4), tflops=round(flop / ms / 1e9, 2),
                 roofline_frac=round(flop / ms / 1e9 / peak, 4), peak_tflops=peak, bound="mfma")


def bench_shared():
    """shared_expert_cpu fp8 (/root/reference/test_moe_fp8.py:87-88) at a DeepSeek-like shape (hidden 7168, width 2048), decode
    and prefill sizes; bytes = the two fp8 weight matrices."""
    N, K = 2048, 7168
    g = torch.Generator(device="cuda").manual_seed(9)
    w1 = ops.convert_weight_packed((torch.randn(2 * N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
    w2 = ops.convert_weight_packed((torch.randn(K, N, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
    s1 = torch.rand(2 * N // 128, K // 128, device="cuda", generator=g) * 1e-4
    s2 = torch.rand(K // 128, N // 128, device="cuda", generator=g) * 1e-4
    for M in (1, 16, 64, 128, 2048):
        hs = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        fo = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        ms = timed(lambda i: ops.shared_expert_cpu(hs, w1, w2, fo, 2.5, False, False, True, s1, s2, [128, 128], None, None, True), 20)
        byts, flop = 3 * N * K, 6 * M * N * K
        emit(op="shared_expert_fp8", M=M, N=N, K=K, ms=round(ms, 4), gbps=round(byts / ms / 1e6, 1), tflops=round(flop / ms / 1e9, 2))
    b1 = ops.convert_weight_packed((torch.randn(2 * N, K, device="cuda", generator=g) * 0.02).bfloat16())
    b2 = ops.convert_weight_packed((torch.randn(K, N, device="cuda", generator=g) * 0.02).bfloat16())
    for M in (1, 64, 128, 2048):
        hs = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        fo = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        ms = timed(lambda i: ops.shared_expert_cpu(hs, b1, b2, fo, 2.5, False, False, False, None, None, None, None, None, True), 20)
        emit(op="shared_expert_bf16", M=M, N=N, K=K, ms=round(ms, 4), gbps=round(6 * N * K / ms / 1e6, 1),
             tflops=round(6 * M * N * K / ms / 1e9, 2))
    i1 = ops.convert_weight_packed(torch.randint(-127, 128, (2 * N, K), device="cuda", generator=g, dtype=torch.int8))
    i2 = ops.convert_weight_packed(torch.randint(-127, 128, (K, N), device="cuda", generator=g, dtype=torch.int8))
    q1 = torch.rand(2 * N, device="cuda", generator=g) * 1e-3
    q2 = torch.rand(K, device="cuda", generator=g) * 1e-3
    for M in (1, 64, 128, 2048):
        hs = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        fo = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        ms = timed(lambda i: ops.shared_expert_cpu(hs, i1, i2, fo, 2.5, False, True, False, q1, q2, None, None, None, True), 20)
        emit(op="shared_expert_int8", M=M, N=N, K=K, ms=round(ms, 4), gbps=round(3 * N * K / ms / 1e6, 1),
             tflops=round(6 * M * N * K / ms / 1e9, 2))


def bench_block():
    """fused_moe_block (router + align in one launch, routed combine folded into the shared expert's last launch) against the
    separate operator calls of the reference flow, decode sizes, device time per call from hipGraph replays."""
    K, N, E, topk, Ns = 2048, 768, 128, 8, 2048
    g = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda *sh: (torch.randn(*sh, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn)
    w1p, w2p = ops.convert_weight_packed(mk(E, 2 * N, K)), ops.convert_weight_packed(mk(E, K, N))
    s1p, s2p = ops.convert_weight_packed(mk(2 * Ns, K)), ops.convert_weight_packed(mk(K, Ns))
    w1s = torch.randn(E, 2 * N // 128, K // 128, device="cuda", generator=g) * 1e-3
    w2s = torch.randn(E, K // 128, N // 128, device="cuda", generator=g) * 1e-3
    s1s = torch.randn(2 * Ns // 128, K // 128, device="cuda", generator=g) * 1e-3
    s2s = torch.randn(K // 128, Ns // 128, device="cuda", generator=g) * 1e-3

    for shared in (False, True):
        for M in (1, 4, 16, 64):
            a = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
            logits = torch.randn(M, E, device="cuda", generator=g).bfloat16()
            sh = (s1p, s2p, s1s, s2s) if shared else (None, None, None, None)

            def fused():
                return ops.fused_moe_block(a, logits, w1p, w2p, topk, True, 1, 1, None, False, False, True, w1s, w2s, [128, 128],
                                           True, *sh, 1.0)

            def separate():
                tw, ids = ops.grouped_topk_cpu(a, logits, topk, True, 1, 1, 0, None, None)
                out = ops.fused_experts_cpu(a, w1p, w2p, tw, ids, False, False, True, w1s, w2s, [128, 128], None, None, True)
                if shared:
                    out = ops.shared_expert_cpu(a, s1p, s2p, out, 1.0, False, False, True, s1s, s2s, [128, 128], None, None, True)
                return out

            t_f, t_s = graph_ms(fused), graph_ms(separate)
            emit(op="fused_moe_block", M=M, shared_expert=shared, ms_block=round(t_f, 4), ms_separate_calls=round(t_s, 4),
                 saved_us=round((t_s - t_f) * 1e3, 2),
                 note="Qwen3-30B-A3B expert dims fp8 (+ a 2048-wide fp8 shared expert); hipGraph replay, device time per call")


def bench_mxfp4():
    """mxfp4_scaled_mm_cpu (/root/reference/test_mxfp4.py): W4A16.  Small M: weights expanded to bf16 in the generic engine's
    loader, bound by the weight bytes (N*K/2 + scales); large M: the fp4 weights as stored on the block-scaled matrix cores
    (gemm_mxfp4.hip), priced against the bf16 roof (two e4m3 terms per activation = the bf16 rate); bmm_cpu rides along
    (test_bmm_fp8.py:131-132)."""
    g = torch.Generator(device="cuda").manual_seed(4)
    for (M, N, K) in ((1, 4096, 4096), (16, 4096, 4096), (128, 4096, 4096), (1024, 12288, 2048), (4096, 4096, 4096)):
        x = (torch.randn(M, K, device="cuda", generator=g) / 10).bfloat16()
        wq = torch.randint(0, 256, (N, K // 2), device="cuda", generator=g, dtype=torch.uint8)
        ws = torch.randint(120, 128, (N, K // 32), device="cuda", generator=g, dtype=torch.uint8)
        sp = ops.convert_scale_packed(ws)
        ms = timed(lambda i: ops.mxfp4_scaled_mm_cpu(x, wq, sp, None, True), 20)
        byts, flop = N * K // 2 + N * K // 32 + 2 * M * (N + K), 2 * M * N * K
        t_hbm, t_mfma = byts / (PEAK_HBM * 1e9), flop / (PEAK_BF16 * 1e12)
        emit(op="mxfp4_scaled_mm", M=M, N=N, K=K, ms=round(ms, 4), tflops=round(flop / ms / 1e9, 2), gbps=round(byts / ms / 1e6, 1),
             bound="hbm" if t_hbm > t_mfma else "mfma", roofline_frac=round(max(t_hbm, t_mfma) * 1e3 / ms, 4))
    for (B, M, N, K) in ((16, 1, 512, 128), (16, 1, 128, 512), (16, 64, 512, 128)):
        a = torch.randn(M, B, K + 64, device="cuda", generator=g).bfloat16().narrow(2, 0, K).transpose(0, 1)
        w = ops.convert_weight_packed(torch.randn(B, N, K, device="cuda", generator=g).bfloat16())
        out = torch.empty(M, B, N + 64, device="cuda", dtype=torch.bfloat16).narrow(2, 0, N).transpose(0, 1)
        ms = timed(lambda i: ops.bmm_cpu(out, a, w, True, None), 20)
        emit(op="bmm_cpu", B=B, M=M, N=N, K=K, ms=round(ms, 4), gbps=round(2 * B * N * K / ms / 1e6, 1), bound="launch")


def bench_attn():
    g = torch.Generator(device="cuda").manual_seed(3)
    dt = torch.bfloat16
    for (B, CTX, HQ, HKV, D, DV) in ((1, 4096, 32, 4, 128, 128), (1, 8192, 16, 2, 128, 128), (4, 2048, 22, 22, 192, 128)):
        T = B * CTX
        q = torch.randn(T, HQ, D, device="cuda", generator=g).to(dt)
        k = torch.randn(T, HKV, D, device="cuda", generator=g).to(dt)
        v = torch.randn(T, HKV, DV, device="cuda", generator=g).to(dt)
        o = torch.empty(T, HQ, DV, device="cuda", dtype=dt)
        rtt = torch.arange(T, device="cuda", dtype=torch.int32).view(B, CTX)
        seq = torch.full((B,), CTX, device="cuda", dtype=torch.int64)
        ext = torch.full((B,), CTX, device="cuda", dtype=torch.int32)
        start = (torch.arange(B, device="cuda", dtype=torch.int32) * CTX)
        ms = timed(lambda i: ops.extend_attention_cpu(q, k, v, o, k, v, rtt, torch.arange(B, device="cuda"), seq, ext, start,
                                                      CTX, 1.0 / D ** 0.5, 0.0), 10)
        flop = B * HQ * (CTX * CTX / 2) * 2 * (D + DV)     # causal
        emit(op="extend_attention", B=B, ctx=CTX, HQ=HQ, HKV=HKV, D=D, DV=DV, ms=round(ms, 4),
             tflops=round(flop / ms / 1e9, 2), roofline_frac=round(flop / ms / 1e9 / PEAK_BF16, 4), bound="mfma")
    # flash_attn_varlen_func: the reference's own bench point (/root/reference/test_flash_attn_varlen.py:117-162:
    # B = 6 sequences of T = 8160, H = Hkv = 6, head dim 72, non-causal) plus causal / GQA / d128 variants
    for (B, T, H, HKV, D, causal) in ((6, 8160, 6, 6, 72, False), (6, 8160, 6, 6, 72, True), (4, 4096, 32, 4, 128, True),
                                      (4, 4096, 32, 4, 64, False)):
        q = torch.randn(B * T, H, D, device="cuda", generator=g).to(dt)
        k = torch.randn(B * T, HKV, D, device="cuda", generator=g).to(dt)
        v = torch.randn(B * T, HKV, D, device="cuda", generator=g).to(dt)
        cu = (torch.arange(B + 1, device="cuda", dtype=torch.int32) * T)
        ms = timed(lambda i: ops.flash_attn_varlen_func(q, k, v, cu, cu, T, T, causal), 10)
        flop = B * H * T * T * 4 * D * (0.5 if causal else 1.0)
        emit(op="flash_attn_varlen_func", B=B, T=T, H=H, HKV=HKV, D=D, causal=causal, ms=round(ms, 4),
             tflops=round(flop / ms / 1e9, 2), roofline_frac=round(flop / ms / 1e9 / PEAK_BF16, 4), bound="mfma")
    for (B, HQ, HKV, D, DV, S, alias) in ((1, 22, 1, 576, 512, 1024, True), (40, 22, 1, 576, 512, 1064, True),
                                            (128, 22, 1, 576, 512, 4096, True), (64, 32, 4, 128, 128, 4096, False)):
        total = B * S
        q = torch.randn(B, HQ, D, device="cuda", generator=g).to(dt)
        kb = torch.randn(total, HKV, D, device="cuda", generator=g).to(dt)
        key = torch.randn(B, HKV, D, device="cuda", generator=g).to(dt)
        vb = kb.narrow(2, 0, DV) if alias else torch.randn(total, HKV, DV, device="cuda", generator=g).to(dt)
        val = key.narrow(2, 0, DV) if alias else torch.randn(B, HKV, DV, device="cuda", generator=g).to(dt)
        o = torch.empty(B, HQ, DV, device="cuda", dtype=dt)
        logits = torch.empty(B, HQ, 8, DV + 1, device="cuda", dtype=torch.float32)
        rtt = torch.arange(total, device="cuda").view(B, S)
        loc = rtt[:, -1].contiguous()
        seq = torch.full((B,), S, device="cuda", dtype=torch.int64)
        ridx = torch.arange(B, device="cuda")
        call = lambda: ops.decode_attention_cpu(q, kb, vb, o, key, val, loc, logits, rtt, ridx, seq, 1.0 / D ** 0.5, 0.0)
        ms = timed(lambda i: call(), 20)
        ms_dev = graph_ms(call)     # the three launches (cache write, split-KV, merge) without the host's launch overhead
        byts = total * HKV * (D if alias else D + DV) * 2
        emit(op="decode_attention", B=B, HQ=HQ, HKV=HKV, D=D, DV=DV, seq=S, v_alias=alias, ms=round(ms, 4),
             ms_device=round(ms_dev, 4), gbps=round(byts / ms_dev / 1e6, 1), roofline_frac=round(byts / ms_dev / 1e6 / PEAK_HBM, 4),
             roofline_frac_eager=round(byts / ms / 1e6 / PEAK_HBM, 4), bound="hbm")


def bench_absorb():
    """qkv_proj_with_rope (/root/reference/test_absorb.py) at decode sizes: HBM-bound on the projection weights."""
    H, R, nope, rope, qlr, hidden = 22, 512, 128, 64, 1536, 7168
    g = torch.Generator(device="cuda").manual_seed(8)
    bf = torch.bfloat16
    qa = ops.convert_weight_packed((torch.randn(qlr, hidden, device="cuda", generator=g) * 0.1).to(bf))
    qb = ops.convert_weight_packed((torch.randn(H * (nope + rope), qlr, device="cuda", generator=g) * 0.1).to(bf))
    kva = ops.convert_weight_packed((torch.randn(R + rope, hidden, device="cuda", generator=g) * 0.1).to(bf))
    wkc = ops.convert_weight_packed((torch.randn(H, R, nope, device="cuda", generator=g) * 0.1).to(bf))
    n1 = torch.randn(qlr, device="cuda", generator=g).to(bf)
    n2 = torch.randn(R, device="cuda", generator=g).to(bf)
    cache = torch.randn(4096, rope, device="cuda", generator=g).to(bf)
    wbytes = 2 * (qlr * hidden + H * (nope + rope) * qlr + (R + rope) * hidden + H * R * nope)
    for B in (1, 16, 128):
        hs = (torch.randn(B, hidden, device="cuda", generator=g) / hidden).to(bf)
        pos = torch.randint(0, 4096, (B,), device="cuda", generator=g)
        call = lambda: ops.qkv_proj_with_rope(hs, qa, qb, kva, wkc, n1, n2, pos, cache, 1e-6, False, False, None, None, None, True, None)
        ms = timed(lambda i: call(), 20)
        try:
            ms_dev = graph_ms(call)          # the ten launches of the one C-ABI call as a hipGraph replay
        except Exception:
            ms_dev = float("nan")
        emit(op="qkv_proj_with_rope_bf16", B=B, hidden=hidden, ms=round(ms, 4), ms_device=round(ms_dev, 4),
             gbps=round(wbytes / ms / 1e6, 1), roofline_frac=round(wbytes / ms / 1e6 / PEAK_HBM, 4), bound="hbm (weights)")


def bench_rows():
    g = torch.Generator(device="cuda").manual_seed(4)
    for rows, two_d in ((128, 22016), (1000, 18432 * 2), (17, 36864)):
        x = torch.randn(rows, two_d, device="cuda", generator=g).bfloat16()
        ms = timed(lambda i: ops.silu_and_mul_cpu(x), 50)
        byts = rows * two_d * 2 * 1.5
        emit(op="silu_and_mul", rows=rows, two_d=two_d, ms=round(ms, 5), gbps=round(byts / ms / 1e6, 1),
             roofline_frac=round(byts / ms / 1e6 / PEAK_HBM, 4), bound="hbm")
    for rows, h in ((1024, 4096), (16384, 2048), (1, 5120)):
        x = torch.randn(rows, h, device="cuda", generator=g).bfloat16()
        r = torch.randn(rows, h, device="cuda", generator=g).bfloat16()
        w = torch.randn(h, device="cuda", generator=g).bfloat16()
        o = torch.empty_like(x)
        ms = timed(lambda i: ops.rmsnorm_cpu(o, x, w, 1e-6), 50)
        emit(op="rmsnorm", rows=rows, hidden=h, ms=round(ms, 5), gbps=round(rows * h * 4 / ms / 1e6, 1),
             roofline_frac=round(rows * h * 4 / ms / 1e6 / PEAK_HBM, 4), bound="hbm")
        ms = timed(lambda i: ops.fused_add_rmsnorm_cpu(x, r, w, 1e-6), 50)
        emit(op="fused_add_rmsnorm", rows=rows, hidden=h, ms=round(ms, 5), gbps=round(rows * h * 8 / ms / 1e6, 1),
             roofline_frac=round(rows * h * 8 / ms / 1e6 / PEAK_HBM, 4), bound="hbm")
    for M, E, G, k, tg in ((16384, 128, 1, 8, 1), (4096, 256, 8, 8, 4)):
        gate = torch.randn(M, E, device="cuda", generator=g).bfloat16()
        ms = timed(lambda i: ops.grouped_topk_cpu(gate, gate, k, True, G, tg, 0, None, None), 50)
        emit(op="grouped_topk", M=M, E=E, G=G, topk=k, ms=round(ms, 5), tokens_per_s=round(M / ms * 1e3))


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    table = {"moe": bench_moe, "moe_literal": bench_moe_literal, "moe_offload": bench_moe_offload, "moe_int8": bench_moe_int8, "gemm": bench_gemm, "shared": bench_shared, "block": bench_block, "mxfp4": bench_mxfp4, "attn": bench_attn, "absorb": bench_absorb, "rows": bench_rows}
    for name, fn in table.items():
        if which in ("all", name):
            fn()
