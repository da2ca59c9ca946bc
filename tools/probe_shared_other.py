"""bf16 / int8 shared_expert_cpu at a DeepSeek-like shape around the decode -> prefill switch (timing probe)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sgl-cpu-tests_amd"))
import sgl_kernel  # noqa: F401,E402

ops = torch.ops.sgl_kernel
g = torch.Generator(device="cuda").manual_seed(9)
N, K = 2048, 7168
PACKED = os.environ.get("ROWMAJOR", "0") != "1"      # ROWMAJOR=1: the weights as the reference's 12-argument call passes them
pack = ops.convert_weight_packed if PACKED else (lambda t: t)
b1 = pack((torch.randn(2 * N, K, device="cuda", generator=g) * 0.02).bfloat16())
b2 = pack((torch.randn(K, N, device="cuda", generator=g) * 0.02).bfloat16())
i1 = pack(torch.randint(-127, 128, (2 * N, K), device="cuda", generator=g, dtype=torch.int8))
i2 = pack(torch.randint(-127, 128, (K, N), device="cuda", generator=g, dtype=torch.int8))
q1 = torch.rand(2 * N, device="cuda", generator=g) * 1e-3
q2 = torch.rand(K, device="cuda", generator=g) * 1e-3


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20


for M in [int(x) for x in sys.argv[1:]]:
    hs = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    fo = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    tb = timed(lambda: ops.shared_expert_cpu(hs, b1, b2, fo, 2.5, False, False, False, None, None, None, None, None, PACKED))
    ti = timed(lambda: ops.shared_expert_cpu(hs, i1, i2, fo, 2.5, False, True, False, q1, q2, None, None, None, PACKED))
    print(json.dumps({"M": M, "bf16_ms": round(tb, 4), "int8_ms": round(ti, 4), "packed": PACKED}), flush=True)
