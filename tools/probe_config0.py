#!/usr/bin/env python3
"""Developer tool: BASELINE.json configs[0] -- the dense GEMMs at (M, N, K) = (128, 4096, 4096) and its neighbours -- as device
time per call (hipGraph replay) for every weight type, beside the eager time tools/bench_ops.py reports and the vendor
library's bf16 GEMM of the same shape (calibration only).  Knobs are read from the environment, so
`SGLK_...=x python tools/probe_config0.py` is an A/B."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from bench_ops import graph_ms, ops, timed, torch  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(2)
shapes = ((128, 4096, 4096),) if os.environ.get("PROBE_ONE") else ((32, 4096, 4096), (64, 4096, 4096), (128, 4096, 4096), (160, 4096, 4096), (128, 2048, 6144), (128, 12288, 2048))
for (M, N, K) in shapes:
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    wb = torch.randn(N, K, device="cuda", generator=g).bfloat16()
    wf = (torch.randn(N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn)
    wi = torch.randint(-127, 127, (N, K), device="cuda", generator=g, dtype=torch.int8)
    sc = torch.randn(N // 128, K // 128, device="cuda", generator=g) * 1e-3
    wfp, wbp, wip = ops.convert_weight_packed(wf), ops.convert_weight_packed(wb), ops.convert_weight_packed(wi)
    si = torch.rand(N, device="cuda", generator=g) * 1e-2
    xq, xs = ops.per_token_quant_int8_cpu(x)
    row = {"M": M, "N": N, "K": K}
    for name, fn, nbytes in (
            ("bf16", lambda: ops.weight_packed_linear(x, wbp, None, True), 2 * N * K),
            ("fp8", lambda: ops.fp8_scaled_mm_cpu(x, wfp, sc, [128, 128], None, torch.bfloat16, True), N * K),
            ("int8", lambda: ops.int8_scaled_mm_cpu(xq, wip, xs, si, None, torch.bfloat16, True), N * K),
            ("int8_with_quant", lambda: ops.int8_scaled_mm_with_quant(x, wip, si, None, torch.bfloat16, True), N * K),
            ("vendor_bf16", lambda: torch.matmul(x, wb.t()), 2 * N * K)):
        dev = graph_ms(fn)
        row[name + "_us"] = round(dev * 1e3, 2)
        row[name + "_gbps"] = round(nbytes / dev / 1e6)
        if name != "vendor_bf16":
            row[name + "_eager_us"] = round(timed(lambda i: fn(), 10) * 1e3, 2)
    print(json.dumps(row), flush=True)
