#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sgl-cpu-tests_amd"))
import torch
import sgl_kernel  # noqa
ops = torch.ops.sgl_kernel
dt = torch.bfloat16
def run(B, CTX, HQ, HKV, D=128, DV=128, iters=20):
    T = B * CTX
    q = torch.randn(T, HQ, D, device="cuda").to(dt); k = torch.randn(T, HKV, D, device="cuda").to(dt); v = torch.randn(T, HKV, DV, device="cuda").to(dt)
    o = torch.empty(T, HQ, DV, device="cuda", dtype=dt)
    rtt = torch.arange(T, device="cuda", dtype=torch.int32).view(B, CTX)
    seq = torch.full((B,), CTX, device="cuda", dtype=torch.int64); ext = torch.full((B,), CTX, device="cuda", dtype=torch.int32)
    start = (torch.arange(B, device="cuda", dtype=torch.int32) * CTX); req = torch.arange(B, device="cuda")
    f = lambda: ops.extend_attention_cpu(q, k, v, o, k, v, rtt, req, seq, ext, start, CTX, 1.0 / D ** 0.5, 0.0)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"B={B} ctx={CTX} HQ={HQ} HKV={HKV}: {ms*1e3:.1f} us")
run(1, 128, 1, 1); run(1, 4096, 1, 1); run(1, 4096, 8, 1); run(1, 4096, 32, 4); run(1, 8192, 1, 1)
