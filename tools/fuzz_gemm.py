"""Randomised parity sweep of the dense GEMM operators (fp8_scaled_mm_cpu, int8_scaled_mm_with_quant, weight_packed_linear) against
oracle/gemm.py (test infrastructure, not a benchmark).  The library picks between weight-streaming split-K kernels, 256-row tile
kernels (with and without split-K) and the generic engine from M, N, K alone -- the shapes are drawn around those crossovers.
usage: python tools/fuzz_gemm.py [iterations] [seed]      -> one line per failure, a summary line at the end"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sgl-cpu-tests_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sgl_kernel  # noqa: F401,E402
import recipes  # noqa: E402
from oracle import gemm as ogemm  # noqa: E402

ops = torch.ops.sgl_kernel
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 90
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
fails = 0
count = {"fp8": 0, "int8": 0, "bf16": 0}


TIMES = os.environ.get("FUZZ_TIME", "0") == "1"     # FUZZ_TIME=1: also time every case and list the ones furthest below their roofline
slow = []


def rate(kind, desc, fn, M, N, K):
    if not TIMES:
        return
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    wb = N * K * (2 if kind == "bf16" else 1)
    t_min = max(2.0 * M * N * K / (5.0e15 if kind == "int8" else 2.5e15), (wb + 2.0 * M * (N + K)) / 8e12) * 1e3
    if t_min > 3e-3:      # problems below ~3 us of roofline time are bound by launch + host latency (~20 us per eager call)
        slow.append((t_min / ms, ms, kind, desc))


def check(kind, desc, out, ref, rms_tol):
    global fails
    o = out.float().cpu()
    k = 2.0 / max(float(ref.abs().max()), 1e-6)      # the reference's atol = rtol = 1e-2 is meant for |out| <~ 2
    ok = torch.allclose((ref * k).bfloat16(), (o * k).bfloat16(), rtol=1e-2, atol=1e-2) and bool(torch.isfinite(o).all())
    rel = float((o - ref).norm() / ref.norm().clamp_min(1e-12))
    if not ok or rel > rms_tol:
        fails += 1
        print(f"FAIL {kind} {desc} rel={rel:.2e} pred={ok}", flush=True)


for it in range(iters):
    kind = ("fp8", "int8", "bf16")[it % 3]
    count[kind] += 1
    M = rng.choice([1, 3, 16, 17, 64, 100, 129, 160, 191, 192, 200, 256, 300, 500, 777, 1000, 1023, 1024, 1025, 1100, 1300, 1500, 1800, 2047, 2048, 2300, 4096])
    N = rng.choice([64, 128, 256, 320, 512, 576, 768, 1024, 1536, 2048, 2304, 4096])
    K = rng.choice([128, 256, 384, 512, 768, 1024, 2048, 2560, 4096, 6144, 7168])
    while M > 1 and 2.0 * M * N * K > 4e10:
        M //= 2
    has_bias = rng.random() < 0.5
    packed = rng.random() < 0.75
    seed = rng.randrange(1 << 30)
    desc = f"it={it} M={M} N={N} K={K} bias={has_bias} packed={packed} seed={seed}"
    if kind == "fp8":
        bn = 64 if N % 128 or rng.random() < 0.5 else 128
        inp = recipes.gemm_fp8_inputs(M, N, K, has_bias, False, seed, bn=bn)
        ref = ogemm.fp8_scaled_mm(inp["data"], inp["w"], inp["scales"], (bn, 128), inp.get("bias"))
        w = inp["w"].cuda()
        w = ops.convert_weight_packed(w) if packed else w
        out = ops.fp8_scaled_mm_cpu(inp["data"].cuda(), w, inp["scales"].cuda(), [bn, 128], inp["bias"].cuda() if has_bias else None,
                                    torch.bfloat16, packed)
        check(kind, desc + f" bn={bn}", out, ref, 4e-3)
        xd, sd, bd = inp["data"].cuda(), inp["scales"].cuda(), inp["bias"].cuda() if has_bias else None
        rate(kind, desc, lambda: ops.fp8_scaled_mm_cpu(xd, w, sd, [bn, 128], bd, torch.bfloat16, packed), M, N, K)
    elif kind == "int8":
        inp = recipes.gemm_int8_inputs(M, N, K, has_bias, seed)
        xq, xs = ogemm.per_token_quant_int8(inp["A"])
        ref = ogemm.int8_scaled_mm(xq, xs, inp["Bq"], inp["Bs"], inp.get("bias"))
        w = inp["Bq"].cuda()
        w = ops.convert_weight_packed(w) if packed else w
        out = ops.int8_scaled_mm_with_quant(inp["A"].cuda(), w, inp["Bs"].cuda(), inp["bias"].cuda() if has_bias else None, torch.bfloat16,
                                            packed)
        check(kind, desc, out, ref, 4e-3)
        xd, sd, bd = inp["A"].cuda(), inp["Bs"].cuda(), inp["bias"].cuda() if has_bias else None
        rate(kind, desc, lambda: ops.int8_scaled_mm_with_quant(xd, w, sd, bd, torch.bfloat16, packed), M, N, K)
    else:
        inp = recipes.gemm_bf16_inputs(M, N, K, has_bias, seed)
        ref = ogemm.linear_bf16(inp["mat1"], inp["mat2"], inp.get("bias"))
        w = inp["mat2"].cuda()
        w = ops.convert_weight_packed(w) if packed else w
        out = ops.weight_packed_linear(inp["mat1"].cuda(), w, inp["bias"].cuda() if has_bias else None, packed)
        check(kind, desc, out, ref, 4e-3)
        xd, bd = inp["mat1"].cuda(), inp["bias"].cuda() if has_bias else None
        rate(kind, desc, lambda: ops.weight_packed_linear(xd, w, bd, packed), M, N, K)
    torch.cuda.synchronize()
for frac, ms, kind, desc in sorted(slow)[:25]:
    print(f"  slow: {frac:.3f} of its roofline, {ms:.4f} ms  {kind} {desc}")
print(f"fuzz_gemm: {count} cases, {fails} failures")
sys.exit(1 if fails else 0)
