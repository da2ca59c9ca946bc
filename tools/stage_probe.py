#!/usr/bin/env python3
"""Developer tool: per-stage times (library HIP events) of one fused_experts configuration.
    python tools/stage_probe.py [fp8|int8|bf16] [M]
SGLK_PROBE_SHAPE=K,N,E,topk picks the shape; SGLK_PROBE_ROTATE=n calls n clones of the weights in turn (a small batch re-reads the same
experts from the Infinity Cache otherwise)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sgl-cpu-tests_amd"))
import torch
import sgl_kernel  # noqa
from sgl_kernel import _lib, _ops
ops = torch.ops.sgl_kernel
kind = sys.argv[1] if len(sys.argv) > 1 else "int8"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
K, N, E, topk = [int(v) for v in os.environ.get("SGLK_PROBE_SHAPE", "2048,768,128,8").split(",")]   # K,N,E,topk
g = torch.Generator(device="cuda").manual_seed(6)
a = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
tw, ids = torch.topk(torch.softmax(torch.randn(M, E, device="cuda", generator=g), dim=-1), topk); ids = ids.to(torch.int32)
if kind == "int8":
    w1 = ops.convert_weight_packed(torch.randint(-127, 128, (E, 2 * N, K), device="cuda", generator=g, dtype=torch.int8))
    w2 = ops.convert_weight_packed(torch.randint(-127, 128, (E, K, N), device="cuda", generator=g, dtype=torch.int8))
    s1 = torch.rand(E, 2 * N, device="cuda", generator=g) * 1e-3; s2 = torch.rand(E, K, device="cuda", generator=g) * 1e-3
    f = lambda: ops.fused_experts_cpu(a, w1, w2, tw, ids, False, True, False, s1, s2, None, None, None, True)
elif kind == "bf16":
    w1 = ops.convert_weight_packed((torch.randn(E, 2 * N, K, device="cuda", generator=g) / K ** 0.5).bfloat16())
    w2 = ops.convert_weight_packed((torch.randn(E, K, N, device="cuda", generator=g) / N ** 0.5).bfloat16())
    s1 = s2 = None
    f = lambda: ops.fused_experts_cpu(a, w1, w2, tw, ids, False, False, False, None, None, None, None, None, True)
else:
    w1 = ops.convert_weight_packed((torch.randn(E, 2 * N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
    w2 = ops.convert_weight_packed((torch.randn(E, K, N, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
    s1 = torch.randn(E, 2 * N // 128, K // 128, device="cuda", generator=g) * 1e-3; s2 = torch.randn(E, K // 128, N // 128, device="cuda", generator=g) * 1e-3
    f = lambda: ops.fused_experts_cpu(a, w1, w2, tw, ids, False, False, True, s1, s2, [128, 128], None, None, True)
ROT = int(os.environ.get("SGLK_PROBE_ROTATE", "1"))
if ROT > 1:
    sets = [(w1, w2)] + [(w1.clone(), w2.clone()) for _ in range(ROT - 1)]
    state = {"i": 0}
    def f():
        state["i"] += 1
        c1, c2 = sets[state["i"] % ROT]
        if kind == "int8":
            return ops.fused_experts_cpu(a, c1, c2, tw, ids, False, True, False, s1, s2, None, None, None, True)
        if kind == "bf16":
            return ops.fused_experts_cpu(a, c1, c2, tw, ids, False, False, False, None, None, None, None, None, True)
        return ops.fused_experts_cpu(a, c1, c2, tw, ids, False, False, True, s1, s2, [128, 128], None, None, True)
L = _lib.lib()
for _ in range(5): f()
torch.cuda.synchronize()
timer = L.sglk_stage_timer_create(64)
_ops.set_stage_timer(timer)
for _ in range(20): f()
torch.cuda.synchronize()
_ops.set_stage_timer(None)
ms = (ctypes.c_float * _lib.NUM_STAGES)(); calls = ctypes.c_int32(0)
L.sglk_stage_timer_read(timer, ms, ctypes.byref(calls))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print(kind, "M", M, "K,N,E,topk", (K, N, E, topk), {n: round(float(ms[i]), 4) for i, n in enumerate(_lib.STAGE_NAMES)}, "sum", round(sum(ms), 4),
      "call (events around 20 calls)", round(e0.elapsed_time(e1) / 20, 4), "path", hex(_ops.last_path), "rotate", ROT)
