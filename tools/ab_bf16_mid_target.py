#!/usr/bin/env python3
"""Developer tool: weight_packed_linear (packed bf16) at 65 ... 1000 rows, device time per call (hipGraph replay), with the
split-K of csrc/gemm_bf16_mid.hip aiming at 512 workgroups (two per CU -- but the 128-row build holds one) or at 256."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from bench_ops import graph_ms, ops, torch  # noqa: E402
from sgl_kernel import _lib  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(4)
for (N, K) in ((4096, 4096), (2048, 6144), (5120, 2048), (12288, 2048)):
    wb = ops.convert_weight_packed(torch.randn(N, K, device="cuda", generator=g).bfloat16())
    for M in (96, 128, 160, 192, 256, 384, 512, 768, 1000):
        x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        row = {"N": N, "K": K, "M": M}
        for t in ("512", "256", "0"):
            os.environ["SGLK_BF16_MID_TARGET"] = t
            _lib.lib().sglk_reload_env()
            row["target%s_us" % t] = round(graph_ms(lambda: ops.weight_packed_linear(x, wb, None, True)) * 1e3, 2)
        print(json.dumps(row), flush=True)
