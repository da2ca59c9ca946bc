"""decode_attention at the reference's MLA shape (B = 40, 22 heads, 1064 keys; /root/reference/test_mla.py:178-183) for a
rocprofv3 --kernel-trace timeline (tools/trace_step.py <dir> kv_cache_write)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sgl-cpu-tests_amd"))
import sgl_kernel  # noqa: F401,E402

ops = torch.ops.sgl_kernel
g = torch.Generator(device="cuda").manual_seed(3)
B, HQ, HKV, D, DV, S = int(os.environ.get("B", 40)), 22, 1, 576, 512, int(os.environ.get("S", 1064))
total = B * S
q = torch.randn(B, HQ, D, device="cuda", generator=g).bfloat16()
kb = torch.randn(total, HKV, D, device="cuda", generator=g).bfloat16()
key = torch.randn(B, HKV, D, device="cuda", generator=g).bfloat16()
vb, val = kb.narrow(2, 0, DV), key.narrow(2, 0, DV)
o = torch.empty(B, HQ, DV, device="cuda", dtype=torch.bfloat16)
logits = torch.empty(B, HQ, 8, DV + 1, device="cuda", dtype=torch.float32)
rtt = torch.arange(total, device="cuda").view(B, S)
loc = rtt[:, -1].contiguous()
seq = torch.full((B,), S, device="cuda", dtype=torch.int64)
ridx = torch.arange(B, device="cuda")
for _ in range(12):
    ops.decode_attention_cpu(q, kb, vb, o, key, val, loc, logits, rtt, ridx, seq, 1.0 / D ** 0.5, 0.0)
torch.cuda.synchronize()
