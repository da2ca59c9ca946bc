#!/usr/bin/env python3
"""How much of a tiny fused_experts call is host time?  (enqueue-only loop vs GPU time by events)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sgl-cpu-tests_amd"))
import torch
import sgl_kernel  # noqa
ops = torch.ops.sgl_kernel
K, N, E, topk = 2048, 768, 128, 8
g = torch.Generator(device="cuda").manual_seed(1)
w1 = ops.convert_weight_packed((torch.randn(E, 2 * N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
w2 = ops.convert_weight_packed((torch.randn(E, K, N, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
w1s = torch.randn(E, 2 * N // 128, K // 128, device="cuda", generator=g) * 1e-3
w2s = torch.randn(E, K // 128, N // 128, device="cuda", generator=g) * 1e-3
for M in (1, 64):
    a = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    tw, ids = torch.topk(torch.softmax(torch.randn(M, E, device="cuda", generator=g), dim=-1), topk); ids = ids.to(torch.int32)
    f = lambda: ops.fused_experts_cpu(a, w1, w2, tw, ids, False, False, True, w1s, w2s, [128, 128], None, None, True)
    for _ in range(20): f()
    torch.cuda.synchronize()
    n = 300
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): f()
    e1.record(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"M={M}: host enqueue {1e6*(t1-t0)/n:.1f} us/call, gpu {1e3*e0.elapsed_time(e1)/n:.1f} us/call, wall {1e6*(t2-t0)/n:.1f} us/call")
