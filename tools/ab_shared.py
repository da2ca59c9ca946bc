"""shared_expert_cpu fp8 at a DeepSeek-like shape between decode and prefill sizes (A/B: SGLK_SHARED_MID_MAX)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sgl-cpu-tests_amd"))
import sgl_kernel  # noqa: F401,E402

ops = torch.ops.sgl_kernel
g = torch.Generator(device="cuda").manual_seed(9)
N, K = 2048, 7168
w1 = ops.convert_weight_packed((torch.randn(2 * N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
w2 = ops.convert_weight_packed((torch.randn(K, N, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
s1 = torch.rand(2 * N // 128, K // 128, device="cuda", generator=g) * 1e-4
s2 = torch.rand(K // 128, N // 128, device="cuda", generator=g) * 1e-4
for M in [int(x) for x in sys.argv[1:]]:
    hs = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    fo = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    fn = lambda: ops.shared_expert_cpu(hs, w1, w2, fo, 2.5, False, False, True, s1, s2, [128, 128], None, None, True)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"M": M, "ms": round(e0.elapsed_time(e1) / 30, 4)}), flush=True)
