import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "sgl-cpu-tests_amd"))
import sgl_kernel
ops = torch.ops.sgl_kernel
g = torch.Generator(device="cuda").manual_seed(2)
for (M, N, K) in ((128, 4096, 4096), (64, 5120, 2048), (16, 4096, 4096)):
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    wb = ops.convert_weight_packed(torch.randn(N, K, device="cuda", generator=g).bfloat16())
    wf = ops.convert_weight_packed((torch.randn(N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
    sc = torch.randn(N // 128, K // 128, device="cuda", generator=g) * 1e-3
    for i in range(20):
        ops.weight_packed_linear(x, wb, None, True)
        ops.fp8_scaled_mm_cpu(x, wf, sc, [128, 128], None, torch.bfloat16, True)
torch.cuda.synchronize()
