#!/usr/bin/env python3
"""A/B the grouped-GEMM tile height (SGLK_MOE_TILE_M=128|256) across M, in ONE process, interleaved rounds."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sgl-cpu-tests_amd"))
import torch
import sgl_kernel  # noqa
ops = torch.ops.sgl_kernel
K, N, E, topk = 2048, 768, 128, 8
TILES = os.environ.get("TILES", "32,128,256").split(",")
g = torch.Generator(device="cuda").manual_seed(1)
w1 = (torch.randn(E, 2 * N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn)
w2 = (torch.randn(E, K, N, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn)
w1s = torch.randn(E, 2 * N // 128, K // 128, device="cuda", generator=g) * 1e-3
w2s = torch.randn(E, K // 128, N // 128, device="cuda", generator=g) * 1e-3
w1p = [ops.convert_weight_packed(w1)]; w2p = [ops.convert_weight_packed(w2)]
w1p.append(w1p[0].clone()); w2p.append(w2p[0].clone())
del w1, w2
for M in [int(x) for x in (sys.argv[1:] or "1 4 16 64 256 512 1024 1536".split())]:
    a = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    score = torch.softmax(torch.randn(M, E, device="cuda", generator=g).bfloat16(), dim=-1, dtype=torch.float32)
    tw, ids = torch.topk(score, topk); ids = ids.to(torch.int32)
    res = {}
    for rnd in range(3):
        for t in TILES:
            os.environ["SGLK_MOE_TILE_M"] = t
            f = lambda i: ops.fused_experts_cpu(a, w1p[i & 1], w2p[i & 1], tw, ids, False, False, True, w1s, w2s, [128, 128], None, None, True)
            for i in range(3): f(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(20): f(i)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(t, []).append(e0.elapsed_time(e1) / 20)
    print(json.dumps({"M": M, **{"ms_tile" + t: round(min(res[t]), 4) for t in TILES}}), flush=True)
