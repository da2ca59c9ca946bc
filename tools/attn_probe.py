"""Two attention launches for counter collection (tools/pmc_cmd.sh): extend at B=1, ctx 4096, 32 heads (FORM 0 kernel) and
flash_attn_varlen at B=4 x 4096 (FORM 1 kernel) -- the same inner loop with little and with plenty of work per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sgl-cpu-tests_amd"))
import sgl_kernel  # noqa: F401,E402

ops = torch.ops.sgl_kernel
g = torch.Generator(device="cuda").manual_seed(3)
dt = torch.bfloat16
B, CTX, HQ, HKV, D = 1, 4096, 32, 4, 128
T = B * CTX
q = torch.randn(T, HQ, D, device="cuda", generator=g).to(dt)
k = torch.randn(T, HKV, D, device="cuda", generator=g).to(dt)
v = torch.randn(T, HKV, D, device="cuda", generator=g).to(dt)
o = torch.empty(T, HQ, D, device="cuda", dtype=dt)
rtt = torch.arange(T, device="cuda", dtype=torch.int32).view(B, CTX)
seq = torch.full((B,), CTX, device="cuda", dtype=torch.int64)
ext = torch.full((B,), CTX, device="cuda", dtype=torch.int32)
start = torch.arange(B, device="cuda", dtype=torch.int32) * CTX
ridx = torch.arange(B, device="cuda")
for _ in range(int(os.environ.get("REPS", "6"))):
    ops.extend_attention_cpu(q, k, v, o, k, v, rtt, ridx, seq, ext, start, CTX, 1.0 / D ** 0.5, 0.0)
B = 4
q = torch.randn(B * CTX, HQ, D, device="cuda", generator=g).to(dt)
k = torch.randn(B * CTX, HKV, D, device="cuda", generator=g).to(dt)
v = torch.randn(B * CTX, HKV, D, device="cuda", generator=g).to(dt)
cu = torch.arange(B + 1, device="cuda", dtype=torch.int32) * CTX
for _ in range(int(os.environ.get("REPS", "6"))):
    ops.flash_attn_varlen_func(q, k, v, cu, cu, CTX, CTX, True)
torch.cuda.synchronize()
