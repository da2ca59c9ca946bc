#!/usr/bin/env python3
"""Developer tool (needs a SGLK_DEV_ABLATE build): in-kernel clock of the GEMM-1 main loop after >= 2 s of load."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sgl-cpu-tests_amd"))
import torch
import sgl_kernel  # noqa
ops = torch.ops.sgl_kernel
K, N, E, topk, M = 2048, 768, 128, 8, 16384
g = torch.Generator(device="cuda").manual_seed(1)
w1 = ops.convert_weight_packed((torch.randn(E, 2 * N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
w2 = ops.convert_weight_packed((torch.randn(E, K, N, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
w1s = torch.randn(E, 2 * N // 128, K // 128, device="cuda", generator=g) * 1e-3
w2s = torch.randn(E, K // 128, N // 128, device="cuda", generator=g) * 1e-3
a = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
tw, ids = torch.topk(torch.softmax(torch.randn(M, E, device="cuda", generator=g), dim=-1), topk); ids = ids.to(torch.int32)
dbg = torch.zeros(32 * 8192, dtype=torch.int64, device="cuda")
os.environ["SGLK_DBG_PTR"] = hex(dbg.data_ptr())
f = lambda: ops.fused_experts_cpu(a, w1, w2, tw, ids, False, False, True, w1s, w2s, [128, 128], None, None, True)
t0 = time.time()
while time.time() - t0 < 2.5:
    for _ in range(50): f()
    torch.cuda.synchronize()
full = dbg.cpu().view(-1, 32).double()
full = full[full[:, 1] > 0]
d = full[:, :2]
print("per-wave DMA-wait cycles (median over tiles):", [int(full[:, 2 + w].median()) for w in range(8)])
print("per-wave barrier-wait cycles (median over tiles):", [int(full[:, 10 + w].median()) for w in range(8)])
clk = d[:, 0] / d[:, 1] * 100e6
print(f"workgroups {len(d)}  in-kernel clock median {clk.median()/1e9:.3f} GHz  (p10 {clk.quantile(0.1)/1e9:.3f}, p90 {clk.quantile(0.9)/1e9:.3f})  loop cycles median {d[:,0].median():.0f}")
