"""Dense GEMMs with tall narrow outputs (N not a multiple of 256, or only a few column tiles) at prefill sizes: timing probe."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sgl-cpu-tests_amd"))
import sgl_kernel  # noqa: F401,E402

ops = torch.ops.sgl_kernel
g = torch.Generator(device="cuda").manual_seed(3)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (M, N, K) in [(2048, 320, 7168), (4096, 128, 6144), (1500, 576, 6144), (2048, 576, 7168), (2048, 512, 4096), (1025, 768, 6144), (4096, 576, 2048)]:
    x = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    wb = ops.convert_weight_packed((torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16())
    wf = ops.convert_weight_packed((torch.randn(N, K, device="cuda", generator=g) * 100).to(torch.float8_e4m3fn))
    sf = torch.rand(N // 64, K // 128, device="cuda", generator=g) * 1e-3
    wi = ops.convert_weight_packed(torch.randint(-127, 128, (N, K), device="cuda", generator=g, dtype=torch.int8))
    si = torch.rand(N, device="cuda", generator=g) * 1e-3
    row = {"M": M, "N": N, "K": K}
    row["bf16_ms"] = round(timed(lambda: ops.weight_packed_linear(x, wb, None, True)), 4)
    row["fp8_ms"] = round(timed(lambda: ops.fp8_scaled_mm_cpu(x, wf, sf, [64, 128], None, torch.bfloat16, True)), 4)
    row["int8_ms"] = round(timed(lambda: ops.int8_scaled_mm_with_quant(x, wi, si, None, torch.bfloat16, True)), 4)
    print(json.dumps(row), flush=True)
