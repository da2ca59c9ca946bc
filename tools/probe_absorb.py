"""qkv_proj_with_rope at one batch size under `rocprofv3 --kernel-trace`: which launches the 0.1 ms are made of."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sgl-cpu-tests_amd"))
import sgl_kernel  # noqa
ops = torch.ops.sgl_kernel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H, R, nope, rope, qlr, hidden = 22, 512, 128, 64, 1536, 7168
g = torch.Generator(device="cuda").manual_seed(8)
bf = torch.bfloat16
qa = ops.convert_weight_packed((torch.randn(qlr, hidden, device="cuda", generator=g) * 0.1).to(bf))
qb = ops.convert_weight_packed((torch.randn(H * (nope + rope), qlr, device="cuda", generator=g) * 0.1).to(bf))
kva = ops.convert_weight_packed((torch.randn(R + rope, hidden, device="cuda", generator=g) * 0.1).to(bf))
wkc = ops.convert_weight_packed((torch.randn(H, R, nope, device="cuda", generator=g) * 0.1).to(bf))
n1 = torch.randn(qlr, device="cuda", generator=g).to(bf)
n2 = torch.randn(R, device="cuda", generator=g).to(bf)
cache = torch.randn(4096, rope, device="cuda", generator=g).to(bf)
hs = (torch.randn(B, hidden, device="cuda", generator=g) / hidden).to(bf)
pos = torch.randint(0, 4096, (B,), device="cuda", generator=g)
for _ in range(20):
    ops.qkv_proj_with_rope(hs, qa, qb, kva, wkc, n1, n2, pos, cache, 1e-6, False, False, None, None, None, True, None)
torch.cuda.synchronize()
