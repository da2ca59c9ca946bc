#!/usr/bin/env python3
"""Developer tool: shared_expert_cpu (fp8 / bf16 / int8, packed weights) across row counts, device time per call (hipGraph replay):
where the dispatch between the weight-streaming and the tile kernels leaves cliffs.  Two shapes: hidden 7168 x width 2048
(DeepSeek-like, /root/reference/test_moe_fp8.py:87-88) and hidden 2048 x width 768."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from bench_ops import graph_ms, ops, torch  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(9)
for (N, K) in ((2048, 7168), (768, 2048)):
    w1 = ops.convert_weight_packed((torch.randn(2 * N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
    w2 = ops.convert_weight_packed((torch.randn(K, N, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
    s1 = torch.rand(2 * N // 128, K // 128, device="cuda", generator=g) * 1e-4
    s2 = torch.rand(K // 128, N // 128, device="cuda", generator=g) * 1e-4
    b1 = ops.convert_weight_packed((torch.randn(2 * N, K, device="cuda", generator=g) / 16).bfloat16())
    b2 = ops.convert_weight_packed((torch.randn(K, N, device="cuda", generator=g) / 16).bfloat16())
    i1 = ops.convert_weight_packed(torch.randint(-127, 128, (2 * N, K), device="cuda", generator=g, dtype=torch.int8))
    i2 = ops.convert_weight_packed(torch.randint(-127, 128, (K, N), device="cuda", generator=g, dtype=torch.int8))
    q1 = torch.rand(2 * N, device="cuda", generator=g) * 1e-3
    q2 = torch.rand(K, device="cuda", generator=g) * 1e-3
    for M in ([1000, 1024, 1280, 1536, 2000, 2048] if os.environ.get("PROBE_BIG") else [1, 16, 64, 128, 129, 160, 192, 256, 257, 384, 512, 768, 1000, 1024, 2048]):
        hs = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        fo = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        row = {"N": N, "K": K, "M": M}
        if os.environ.get("PROBE_I8"):
            row["int8_us"] = round(graph_ms(lambda: ops.shared_expert_cpu(hs, i1, i2, fo, 2.5, False, True, False, q1, q2, None, None, None, True)) * 1e3, 2)
            print(json.dumps(row), flush=True)
            continue
        row["fp8_us"] = round(graph_ms(lambda: ops.shared_expert_cpu(hs, w1, w2, fo, 2.5, False, False, True, s1, s2, [128, 128], None, None, True)) * 1e3, 2)
        row["bf16_us"] = round(graph_ms(lambda: ops.shared_expert_cpu(hs, b1, b2, fo, 2.5, False, False, False, None, None, None, None, None, True)) * 1e3, 2)
        row["int8_us"] = round(graph_ms(lambda: ops.shared_expert_cpu(hs, i1, i2, fo, 2.5, False, True, False, q1, q2, None, None, None, True)) * 1e3, 2)
        print(json.dumps(row), flush=True)
