#!/usr/bin/env python3
"""Developer tool: fp8_scaled_mm_cpu (packed fp8 weights) at decode sizes, replayed on one weight and on rotating weight copies
(cold caches), for A/Bs by knob:  SGLK_W_NT=0|1 python tools/dense_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from bench_ops import graph_ms, ops, torch  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(5)
for (N, K) in ((4096, 4096), (2048, 7168), (12288, 2048)):
    copies = max(2, int(600e6 // (N * K)) + 1)
    ws = [ops.convert_weight_packed((torch.randn(N, K, device="cuda", generator=g) * 100).clamp(-400, 400).to(torch.float8_e4m3fn))
          for _ in range(copies)]
    sc = torch.rand(N // 128, K // 128, device="cuda", generator=g) * 1e-2
    for M in (1, 16, 64, 96):
        x = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        one = graph_ms(lambda: ops.fp8_scaled_mm_cpu(x, ws[0], sc, [128, 128], None, torch.bfloat16, True))
        st = {"i": 0}
        def rot():
            st["i"] += 1
            return ops.fp8_scaled_mm_cpu(x, ws[st["i"] % copies], sc, [128, 128], None, torch.bfloat16, True)
        cold = graph_ms(rot)
        print(json.dumps({"M": M, "N": N, "K": K, "nt": os.environ.get("SGLK_W_NT", "rule"), "us_replay": round(one * 1e3, 2),
                          "us_rotating": round(cold * 1e3, 2), "gbps_rotating": round(N * K / cold / 1e6, 1)}), flush=True)
