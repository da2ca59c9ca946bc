#!/usr/bin/env python3
"""Developer tool: int8_scaled_mm_cpu (packed weights) between 129 and 1000 rows, device time per call (hipGraph replay):
the 256-row tile kernel (SGLK_I8_DENSE_MID_WGS=0) against the weight-streaming kernel with several 128-row tiles (=1000: always)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from bench_ops import graph_ms, ops, torch  # noqa: E402
from sgl_kernel import _lib  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(3)
for (N, K) in ((4096, 4096), (2048, 6144), (5120, 2048), (12288, 2048), (2048, 4096)):
    wi = ops.convert_weight_packed(torch.randint(-127, 127, (N, K), device="cuda", generator=g, dtype=torch.int8))
    si = torch.rand(N, device="cuda", generator=g) * 1e-2
    for M in (129, 160, 192, 256, 384, 512, 768, 1000):
        x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        xq, xs = ops.per_token_quant_int8_cpu(x)
        row = {"N": N, "K": K, "M": M, "wgs256": -(-M // 256) * (N // 256)}
        for name, v in (("tile256", "0"), ("mid", "1000")):
            os.environ["SGLK_I8_DENSE_MID_WGS"] = v
            _lib.lib().sglk_reload_env()
            row[name + "_us"] = round(graph_ms(lambda: ops.int8_scaled_mm_cpu(xq, wi, xs, si, None, torch.bfloat16, True)) * 1e3, 2)
        print(json.dumps(row), flush=True)
