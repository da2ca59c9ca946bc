"""int8 / bf16 fused_experts at the Qwen3 expert shape for a list of batch sizes (crossover A/B: SGLK_MID_I8_HI / SGLK_MID_BF16_HI)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sgl-cpu-tests_amd"))
import sgl_kernel  # noqa: F401,E402

ops = torch.ops.sgl_kernel
K, N, E, topk = 2048, 768, 128, 8
g = torch.Generator(device="cuda").manual_seed(6)
w1 = ops.convert_weight_packed(torch.randint(-127, 128, (E, 2 * N, K), device="cuda", generator=g, dtype=torch.int8))
w2 = ops.convert_weight_packed(torch.randint(-127, 128, (E, K, N), device="cuda", generator=g, dtype=torch.int8))
w1s = torch.rand(E, 2 * N, device="cuda", generator=g) * 1e-3
w2s = torch.rand(E, K, device="cuda", generator=g) * 1e-3
b1 = ops.convert_weight_packed((torch.randn(E, 2 * N, K, device="cuda", generator=g) * 0.02).bfloat16())
b2 = ops.convert_weight_packed((torch.randn(E, K, N, device="cuda", generator=g) * 0.02).bfloat16())


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for M in [int(x) for x in sys.argv[1:]]:
    a = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    score = torch.softmax(torch.randn(M, E, device="cuda", generator=g).bfloat16(), dim=-1, dtype=torch.float32)
    tw, ids = torch.topk(score, topk)
    ids = ids.to(torch.int32)
    i8 = timed(lambda: ops.fused_experts_cpu(a, w1, w2, tw, ids, False, True, False, w1s, w2s, None, None, None, True))
    bf = timed(lambda: ops.fused_experts_cpu(a, b1, b2, tw, ids, False, False, False, None, None, None, None, None, True))
    print(json.dumps({"M": M, "int8_ms": round(i8, 4), "bf16_ms": round(bf, 4)}), flush=True)
