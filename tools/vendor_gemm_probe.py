#!/usr/bin/env python3
"""Calibration only (not a product path): what the vendor library behind torch.matmul reaches for plain bf16 GEMMs of the
shapes tools/bench_ops.py measures, on the same box.  random data (the chip's clock under load depends on the data)."""
import torch
shapes = ((1000, 18432, 2560), (4096, 1536, 2048), (4096, 12288, 2048), (4096, 2048, 6144), (16384, 1536, 2048), (8192, 8192, 8192))
for (M, N, K) in shapes:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    for _ in range(3): y = x @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = x @ w.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"torch.matmul bf16 ({M},{N},{K}): {ms:.4f} ms  {2*M*N*K/ms/1e9:.1f} TFLOP/s")
