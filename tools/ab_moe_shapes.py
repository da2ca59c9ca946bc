"""fp8 fused_experts at arbitrary expert shapes: the mid (96-row weight-streaming) kernel against the 256-row tile kernel around
their crossover (average rows per expert), A/B by SGLK_MOE_TILE_M inside one process (sglk_reload_env).
usage: python tools/ab_moe_shapes.py N K E rows_per_expert [rows_per_expert ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sgl-cpu-tests_amd"))
import sgl_kernel  # noqa: F401,E402
from sgl_kernel import _lib  # noqa: E402

ops = torch.ops.sgl_kernel
N, K, E = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
topk = 8
g = torch.Generator(device="cuda").manual_seed(5)
w1 = ops.convert_weight_packed((torch.randn(E, 2 * N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
w2 = ops.convert_weight_packed((torch.randn(E, K, N, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
s1 = torch.rand(E, 2 * N // 128, K // 128, device="cuda", generator=g) * 1e-4
s2 = torch.rand(E, K // 128, N // 128, device="cuda", generator=g) * 1e-4


def timed(fn, iters=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for rpe in [int(x) for x in sys.argv[4:]]:
    M = rpe * E // topk
    a = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    score = torch.softmax(torch.randn(M, E, device="cuda", generator=g).bfloat16(), dim=-1, dtype=torch.float32)
    tw, ids = torch.topk(score, topk)
    ids = ids.to(torch.int32)
    row = {"N": N, "K": K, "E": E, "rows_per_expert": rpe, "M": M}
    for name, v in (("default", None), ("mid96", "96"), ("t256", "256")):
        if v is None:
            os.environ.pop("SGLK_MOE_TILE_M", None)
        else:
            os.environ["SGLK_MOE_TILE_M"] = v
        _lib.lib().sglk_reload_env()
        row[name + "_ms"] = round(timed(lambda: ops.fused_experts_cpu(a, w1, w2, tw, ids, False, False, True, s1, s2, [128, 128], None, None, True)), 4)
    print(json.dumps(row), flush=True)
