#!/usr/bin/env python3
"""Developer tool: device time of decode_attention_cpu at one shape, for A/Bs of the split-KV kernel.

    SGLK_DEC_SHAPE=B,HQ,HKV,D,DV,seq,alias python tools/decode_probe.py        (default: 128,22,1,576,512,4096,1)

With a SGLK_DEV_ABLATE library (SGLK_LIB_PATH=.../libsglk_dev.so) SGLK_RESCALE=1 drops the tile arithmetic and SGLK_RESCALE=2 the
row requests after the prologue (WRONG results, timing only): what the memory stream alone and the arithmetic alone cost.
SGLK_DEC_PERM=1 draws the cache rows as a random permutation instead of consecutive rows."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from bench_ops import graph_ms, ops, timed, torch  # noqa: E402

B, HQ, HKV, D, DV, S, alias = (int(x) for x in os.environ.get("SGLK_DEC_SHAPE", "128,22,1,576,512,4096,1").split(","))
g = torch.Generator(device="cuda").manual_seed(3)
dt = torch.bfloat16
total = B * S
q = torch.randn(B, HQ, D, device="cuda", generator=g).to(dt)
kb = torch.randn(total, HKV, D, device="cuda", generator=g).to(dt)
key = torch.randn(B, HKV, D, device="cuda", generator=g).to(dt)
vb = kb.narrow(2, 0, DV) if alias else torch.randn(total, HKV, DV, device="cuda", generator=g).to(dt)
val = key.narrow(2, 0, DV) if alias else torch.randn(B, HKV, DV, device="cuda", generator=g).to(dt)
o = torch.empty(B, HQ, DV, device="cuda", dtype=dt)
logits = torch.empty(B, HQ, 8, DV + 1, device="cuda", dtype=torch.float32)
if os.environ.get("SGLK_DEC_PERM"):
    rtt = torch.randperm(total, device="cuda", generator=g).view(B, S)
else:
    rtt = torch.arange(total, device="cuda").view(B, S)
loc = rtt[:, -1].contiguous()
seq = torch.full((B,), S, device="cuda", dtype=torch.int64)
ridx = torch.arange(B, device="cuda")
call = lambda: ops.decode_attention_cpu(q, kb, vb, o, key, val, loc, logits, rtt, ridx, seq, 1.0 / D ** 0.5, 0.0)
ms = timed(lambda i: call(), 20)
ms_dev = graph_ms(call)
byts = total * HKV * (D if alias else D + DV) * 2
print(json.dumps({"shape": [B, HQ, HKV, D, DV, S, alias], "abl": os.environ.get("SGLK_RESCALE", ""), "perm": bool(os.environ.get("SGLK_DEC_PERM")),
                  "ms": round(ms, 4), "ms_device": round(ms_dev, 4), "gbps": round(byts / ms_dev / 1e6, 1)}), flush=True)
