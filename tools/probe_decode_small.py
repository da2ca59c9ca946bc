"""decode_attention_cpu at small / short shapes: device time per kernel (torch profiler) beside the eager time."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sgl-cpu-tests_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import sgl_kernel  # noqa: F401,E402
import recipes  # noqa: E402

ops = torch.ops.sgl_kernel
CASES = [(40, 22, 22, 192, 128, 33, False), (40, 32, 4, 128, 128, 200, False), (5, 22, 1, 576, 512, 3000, True), (40, 40, 8, 128, 128, 200, False),
         (17, 22, 22, 192, 128, 200, False), (1, 40, 8, 128, 128, 1024, False), (40, 22, 1, 576, 512, 1064, True)]
for B, HQ, HKV, D, DV, S, alias in CASES:
    inp = recipes.decode_inputs(B, HQ, HKV, D, DV, S, alias, 5)
    kb, key = inp["k_buffer"].cuda(), inp["key"].cuda()
    vb, value = (kb.narrow(2, 0, DV), key.narrow(2, 0, DV)) if alias else (inp["v_buffer"].cuda(), inp["value"].cuda())
    o = torch.empty(B, HQ, DV, dtype=torch.bfloat16, device="cuda")
    logits = torch.empty(B, HQ, 8, DV + 1, dtype=torch.float32, device="cuda")
    q, loc, rtt, rid, sl = inp["q"].cuda(), inp["loc"].cuda(), inp["req_to_token"].cuda(), inp["b_req_idx"].cuda(), inp["b_seq_len"].cuda()
    fn = lambda: ops.decode_attention_cpu(q, kb, vb, o, key, value, loc, logits, rtt, rid, sl, 1.0 / D ** 0.5, 0.0)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    eager = e0.elapsed_time(e1) / 10
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    ks = [(e.key.split("(")[0][-44:], round(e.device_time_total / e.count, 1)) for e in prof.key_averages()]
    print(f"B={B} HQ={HQ}/{HKV} D={D}/{DV} S={S}: eager {eager * 1e3:.1f} us, kernels {ks}", flush=True)
