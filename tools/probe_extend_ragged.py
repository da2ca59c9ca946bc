"""extend_attention_cpu on ragged sequences with a paged prefix: kernel time by the torch profiler (device side), eager time beside it."""
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
CASES = [(2, 2100, 22, 22, 128, 128, False, 11), (2, 2100, 8, 2, 192, 128, False, 12), (3, 2100, 22, 22, 64, 64, False, 13),
         (1, 2100, 22, 22, 128, 96, False, 14), (5, 2100, 32, 4, 128, 128, False, 15), (1, 4096, 32, 4, 128, 128, False, 16)]
for B, N_CTX, HQ, HKV, D, DV, mla, seed in CASES:
    inp = recipes.extend_inputs(B, N_CTX, HQ, HKV, D, DV, mla, seed) if N_CTX != 4096 else recipes.extend_inputs_fixed(B, N_CTX, HQ, HKV, D, DV, mla, seed)
    d = {k: v.cuda() for k, v in inp.items()}
    o = torch.empty(inp["q_extend"].shape[0], HQ, DV, dtype=torch.bfloat16, device="cuda")
    mx = int(inp["b_extend"].max())
    fn = lambda: ops.extend_attention_cpu(d["q_extend"], d["k_extend"], d["v_extend"], o, d["k_buffer"], d["v_buffer"], d["req_to_tokens"],
                                          d["b_req_idx"], d["b_seq_len"], d["b_extend"], d["b_start_loc_extend"], mx, 1.0 / D ** 0.5, 0.0)
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
    ks = [(e.key[:60], e.device_time_total / e.count) for e in prof.key_averages() if "attn" in e.key or "extend" in e.key]
    fl = sum(2.0 * HQ * int(e) * (int(p) + (int(e) + 1) / 2.0) * (D + DV) for e, p in zip(inp["b_extend"], inp["b_seq_len"] - inp["b_extend"]))
    print(f"B={B} HQ={HQ}/{HKV} D={D}/{DV} ext={inp['b_extend'].tolist()} prefix={(inp['b_seq_len'] - inp['b_extend']).tolist()}: eager {eager * 1e3:.1f} us, "
          f"kernels {[(k, round(t, 1)) for k, t in ks]} us, {fl / 1e9:.2f} GFLOP", flush=True)
