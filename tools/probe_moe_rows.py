#!/usr/bin/env python3
"""Developer tool: fused_experts_cpu at Qwen3-30B-A3B expert dims (E = 128, top-8, N = 768, K = 2048) for the three weight types across
token counts, device time per call (hipGraph replay): where the switch between the weight-streaming and the tile kernels leaves steps."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from bench_ops import graph_ms, ops, torch  # noqa: E402

E, topk, N, K = 128, 8, 768, 2048
g = torch.Generator(device="cuda").manual_seed(21)
w1f = ops.convert_weight_packed((torch.randn(E, 2 * N, K, device="cuda", generator=g) * 100).clamp(-400, 400).to(torch.float8_e4m3fn))
w2f = ops.convert_weight_packed((torch.randn(E, K, N, device="cuda", generator=g) * 100).clamp(-400, 400).to(torch.float8_e4m3fn))
s1 = torch.rand(E, 2 * N // 128, K // 128, device="cuda", generator=g) * 1e-2
s2 = torch.rand(E, K // 128, N // 128, device="cuda", generator=g) * 1e-2
w1b = ops.convert_weight_packed((torch.randn(E, 2 * N, K, device="cuda", generator=g) / 16).bfloat16())
w2b = ops.convert_weight_packed((torch.randn(E, K, N, device="cuda", generator=g) / 16).bfloat16())
w1i = ops.convert_weight_packed(torch.randint(-127, 128, (E, 2 * N, K), device="cuda", generator=g, dtype=torch.int8))
w2i = ops.convert_weight_packed(torch.randint(-127, 128, (E, K, N), device="cuda", generator=g, dtype=torch.int8))
q1 = torch.rand(E, 2 * N, device="cuda", generator=g) * 1e-3
q2 = torch.rand(E, K, device="cuda", generator=g) * 1e-3
for M in (16, 64, 128, 256, 384, 512, 640, 768, 896, 1024, 1280, 1536, 2048, 3072, 4096):
    a = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    score = torch.softmax(torch.randn(M, E, device="cuda", generator=g).bfloat16(), dim=-1, dtype=torch.float32)
    tw, ids = torch.topk(score, topk)
    ids = ids.to(torch.int32)
    row = {"M": M}
    row["fp8_us"] = round(graph_ms(lambda: ops.fused_experts_cpu(a, w1f, w2f, tw, ids, False, False, True, s1, s2, [128, 128], None, None, True)) * 1e3, 1)
    row["int8_us"] = round(graph_ms(lambda: ops.fused_experts_cpu(a, w1i, w2i, tw, ids, False, True, False, q1, q2, None, None, None, True)) * 1e3, 1)
    row["bf16_us"] = round(graph_ms(lambda: ops.fused_experts_cpu(a, w1b, w2b, tw, ids, False, False, False, None, None, None, None, None, True)) * 1e3, 1)
    print(json.dumps(row), flush=True)
