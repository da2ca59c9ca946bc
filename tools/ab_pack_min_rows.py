#!/usr/bin/env python3
"""Developer tool: dense GEMMs called with ROW-MAJOR weights (is_vnni=False, what the reference's tests pass) below 192 rows: the
generic engine (SGLK_PACK_MIN_ROWS=192) against re-tiling the weight into the workspace first and running the packed kernels (=1).
Device time per call in us (hipGraph replay)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from bench_ops import graph_ms, ops, torch  # noqa: E402
from sgl_kernel import _lib  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(12)
for (N, K) in ((4096, 4096), (2048, 7168), (12288, 2048), (512, 1024)):
    wb = torch.randn(N, K, device="cuda", generator=g).bfloat16()
    wf = (torch.randn(N, K, device="cuda", generator=g) * 100).clamp(-400, 400).to(torch.float8_e4m3fn)
    sc = torch.rand(N // 128, K // 128, device="cuda", generator=g) * 1e-2
    wi = torch.randint(-127, 127, (N, K), device="cuda", generator=g, dtype=torch.int8)
    si = torch.rand(N, device="cuda", generator=g) * 1e-2
    for M in (1, 4, 16, 32, 64, 128, 191):
        x = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        row = {"N": N, "K": K, "M": M}
        for name, v in (("generic", "192"), ("repack", "1")):
            os.environ["SGLK_PACK_MIN_ROWS"] = v
            _lib.lib().sglk_reload_env()
            row["bf16_" + name] = round(graph_ms(lambda: ops.weight_packed_linear(x, wb, None, False)) * 1e3, 2)
            row["fp8_" + name] = round(graph_ms(lambda: ops.fp8_scaled_mm_cpu(x, wf, sc, [128, 128], None, torch.bfloat16, False)) * 1e3, 2)
            row["i8_" + name] = round(graph_ms(lambda: ops.int8_scaled_mm_with_quant(x, wi, si, None, torch.bfloat16, False)) * 1e3, 2)
        print(json.dumps(row), flush=True)
