"""Per-stage HIP-event times of fp8 fused_experts (align / GEMM-1 / GEMM-2 / combine) over an M sweep at the Qwen3-30B-A3B
expert shape - shows which stage a mid-size batch spends its time in.  Usage: python tools/moe_stage_sweep.py [M ...]"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sgl-cpu-tests_amd"))
import sgl_kernel  # noqa: E402,F401
from sgl_kernel import _lib, _ops  # noqa: E402

ops = torch.ops.sgl_kernel


def main():
    Ms = [int(x) for x in sys.argv[1:]] or [256, 512, 1024, 2048, 3929, 4096, 8192]
    K, N, E, topk = 2048, 768, 128, 8
    g = torch.Generator(device="cuda").manual_seed(1)
    w1 = ops.convert_weight_packed((torch.randn(E, 2 * N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
    w2 = ops.convert_weight_packed((torch.randn(E, K, N, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
    w1b, w2b = w1.clone(), w2.clone()
    w1s = torch.randn(E, 2 * N // 128, K // 128, device="cuda", generator=g) * 1e-3
    w2s = torch.randn(E, K // 128, N // 128, device="cuda", generator=g) * 1e-3
    L = _lib.lib()
    iters = 30
    timer = L.sglk_stage_timer_create(iters + 8)
    for M in Ms:
        a = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
        score = torch.softmax(torch.randn(M, E, device="cuda", generator=g).bfloat16(), dim=-1, dtype=torch.float32)
        tw, ids = torch.topk(score, topk)
        ids = ids.to(torch.int32)
        call = lambda i: ops.fused_experts_cpu(a, (w1, w1b)[i & 1], (w2, w2b)[i & 1], tw, ids, False, False, True, w1s, w2s,
                                               [128, 128], None, None, True)
        for i in range(5):
            call(i)
        torch.cuda.synchronize()
        _ops.set_stage_timer(timer)
        L.sglk_stage_timer_reset(timer)
        for i in range(iters):
            call(i)
        torch.cuda.synchronize()
        _ops.set_stage_timer(None)
        ms = (ctypes.c_float * _lib.NUM_STAGES)()
        calls = ctypes.c_int32(0)
        _lib.check(L.sglk_stage_timer_read(timer, ms, ctypes.byref(calls)), "stage_timer_read")
        st = {n: round(float(ms[i]), 4) for i, n in enumerate(_lib.STAGE_NAMES)}
        counts = torch.bincount(ids.flatten().long(), minlength=E)
        print(json.dumps({"M": M, "stage_ms": st, "sum_ms": round(sum(st.values()), 4), "rows_per_expert_avg": M * topk / E,
                          "rows_max": int(counts.max()), "tiles256": int(((counts + 255) // 256).sum()),
                          "tiles128": int(((counts + 127) // 128).sum())}), flush=True)
    L.sglk_stage_timer_destroy(timer)


if __name__ == "__main__":
    main()
