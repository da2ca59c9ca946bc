#!/usr/bin/env python3
"""Developer tool: fp8_scaled_mm_cpu / int8_scaled_mm_cpu (packed weights) at 16 ... 1000 rows, device time per call (hipGraph
replay), split-K of the weight-streaming kernels by SGLK_MID_DENSE_MODEL = 0 (aim at 512 workgroups) / 1 (rounds model, two
workgroups per CU) / 2 (rounds model, one per CU)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from bench_ops import graph_ms, ops, torch  # noqa: E402
from sgl_kernel import _lib  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(6)
for (N, K) in ((4096, 4096), (2048, 6144), (5120, 2048), (12288, 2048), (2048, 7168)):
    wf = ops.convert_weight_packed((torch.randn(N, K, device="cuda", generator=g) * 100).clamp(-400, 400).to(torch.float8_e4m3fn))
    sc = torch.rand(N // 128, K // 128, device="cuda", generator=g) * 1e-2
    wi = ops.convert_weight_packed(torch.randint(-127, 127, (N, K), device="cuda", generator=g, dtype=torch.int8))
    si = torch.rand(N, device="cuda", generator=g) * 1e-2
    for M in (1, 16, 64, 128, 160, 256, 384, 512, 1000):
        x = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        xq, xs = ops.per_token_quant_int8_cpu(x)
        row = {"N": N, "K": K, "M": M}
        for m in ("0", "2"):
            os.environ["SGLK_MID_DENSE_MODEL"] = m
            _lib.lib().sglk_reload_env()
            row["fp8_m%s" % m] = round(graph_ms(lambda: ops.fp8_scaled_mm_cpu(x, wf, sc, [128, 128], None, torch.bfloat16, True)) * 1e3, 2)
            row["i8_m%s" % m] = round(graph_ms(lambda: ops.int8_scaled_mm_cpu(xq, wi, xs, si, None, torch.bfloat16, True)) * 1e3, 2)
        print(json.dumps(row), flush=True)
