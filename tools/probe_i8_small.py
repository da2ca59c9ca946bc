import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sgl-cpu-tests_amd")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgl_kernel, recipes
ops = torch.ops.sgl_kernel
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, N, K) in [(1023, 128, 384), (1023, 128, 512), (1024, 128, 384), (1023, 256, 384), (500, 128, 384), (1023, 64, 384), (2048, 128, 384)]:
    inp = recipes.gemm_int8_inputs(M, N, K, True, 1)
    A, Bq, Bs, bias = inp["A"].cuda(), inp["Bq"].cuda(), inp["Bs"].cuda(), inp["bias"].cuda()
    Bp = ops.convert_weight_packed(Bq)
    r = {}
    r["rowmajor"] = t(lambda: ops.int8_scaled_mm_with_quant(A, Bq, Bs, bias, torch.bfloat16, False))
    r["packed"] = t(lambda: ops.int8_scaled_mm_with_quant(A, Bp, Bs, bias, torch.bfloat16, True))
    print(M, N, K, {k: round(v, 4) for k, v in r.items()}, flush=True)
