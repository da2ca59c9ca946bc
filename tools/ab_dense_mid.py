"""bf16 / fp8 / int8 dense GEMMs with packed weights around the weight-streaming -> 256-row crossover (A/B: SGLK_DENSE_MID_MAX)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sgl-cpu-tests_amd"))
import sgl_kernel  # noqa: F401,E402

ops = torch.ops.sgl_kernel
g = torch.Generator(device="cuda").manual_seed(2)


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for (N, K) in ((4096, 4096), (12288, 2048), (2048, 6144)):
    wb = ops.convert_weight_packed((torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16())
    w8 = ops.convert_weight_packed((torch.randn(N, K, device="cuda", generator=g) * 100).clamp(-400, 400).to(torch.float8_e4m3fn))
    s8 = torch.rand(N // 128, K // 128, device="cuda", generator=g) * 1e-3
    wi = ops.convert_weight_packed(torch.randint(-127, 128, (N, K), device="cuda", generator=g, dtype=torch.int8))
    si = torch.rand(N, device="cuda", generator=g) * 1e-3
    for M in [int(x) for x in sys.argv[1:]]:
        x = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        tb = timed(lambda: ops.weight_packed_linear(x, wb, None, True))
        t8 = timed(lambda: ops.fp8_scaled_mm_cpu(x, w8, s8, [128, 128], None, torch.bfloat16, True))
        ti = timed(lambda: ops.int8_scaled_mm_with_quant(x, wi, si, None, torch.bfloat16, True))
        print(json.dumps({"N": N, "K": K, "M": M, "bf16_ms": round(tb, 4), "fp8_ms": round(t8, 4), "int8_ms": round(ti, 4)}), flush=True)
