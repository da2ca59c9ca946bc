import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "sgl-cpu-tests_amd"))
import sgl_kernel
ops = torch.ops.sgl_kernel
K, N, E, topk = 2048, 768, 128, 8
g = torch.Generator(device="cuda").manual_seed(6)
b1 = ops.convert_weight_packed((torch.randn(E, 2 * N, K, device="cuda", generator=g) * 0.02).bfloat16())
b2 = ops.convert_weight_packed((torch.randn(E, K, N, device="cuda", generator=g) * 0.02).bfloat16())
for M in (1, 16, 64, 256, 512):
    a = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    score = torch.softmax(torch.randn(M, E, device="cuda", generator=g).bfloat16(), dim=-1, dtype=torch.float32)
    tw, ids = torch.topk(score, topk); ids = ids.to(torch.int32)
    f = lambda: ops.fused_experts_cpu(a, b1, b2, tw, ids, False, False, False, None, None, None, None, None, True)
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30): f()
    e.record(); torch.cuda.synchronize()
    print(os.environ.get("SGLK_NO_BF16_MID", "mid"), M, round(s.elapsed_time(e) / 30, 4), flush=True)
