#!/usr/bin/env python3
"""Developer tool: dense GEMMs (packed weights) at 192 ... 1000 rows on the weight-streaming kernels (thresholds forced up) against
the 256-row tile kernels (thresholds 0), device time per call (hipGraph replay); wgs256 = workgroups of the tile kernel."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from bench_ops import graph_ms, ops, torch  # noqa: E402
from sgl_kernel import _lib  # noqa: E402

KN = ("SGLK_DENSE_MID_WGS_BF16", "SGLK_DENSE_MID_WGS_FP8", "SGLK_I8_DENSE_MID_WGS")
g = torch.Generator(device="cuda").manual_seed(8)
for (N, K) in ((4096, 4096), (2048, 6144), (5120, 2048), (12288, 2048), (2048, 7168), (8192, 7168), (18432, 2560)):
    wb = ops.convert_weight_packed(torch.randn(N, K, device="cuda", generator=g).bfloat16())
    wf = ops.convert_weight_packed((torch.randn(N, K, device="cuda", generator=g) * 100).clamp(-400, 400).to(torch.float8_e4m3fn))
    sc = torch.rand(N // 128, K // 128, device="cuda", generator=g) * 1e-2
    wi = ops.convert_weight_packed(torch.randint(-127, 127, (N, K), device="cuda", generator=g, dtype=torch.int8))
    si = torch.rand(N, device="cuda", generator=g) * 1e-2
    for M in ((1024, 1280, 1536, 2000) if os.environ.get("PROBE_BIG") else (192, 256, 384, 512, 768, 1000)):
        x = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        xq, xs = ops.per_token_quant_int8_cpu(x)
        row = {"N": N, "K": K, "M": M, "wgs256": -(-M // 256) * (N // 256)}
        for name, v in (("tile", "0"), ("mid", "100000")):
            for k in KN:
                os.environ[k] = v
            if v == "0":
                os.environ["SGLK_I8_DENSE_MID_WGS"] = "1"    # 0 would also move 129 ... 191 rows; not swept here
            _lib.lib().sglk_reload_env()
            row["bf16_" + name] = round(graph_ms(lambda: ops.weight_packed_linear(x, wb, None, True)) * 1e3, 2)
            row["fp8_" + name] = round(graph_ms(lambda: ops.fp8_scaled_mm_cpu(x, wf, sc, [128, 128], None, torch.bfloat16, True)) * 1e3, 2)
            row["i8_" + name] = round(graph_ms(lambda: ops.int8_scaled_mm_cpu(xq, wi, xs, si, None, torch.bfloat16, True)) * 1e3, 2)
        print(json.dumps(row), flush=True)
