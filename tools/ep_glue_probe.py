"""Where an expert-parallel step spends its time besides the exchange itself: runs ExpertParallelMoE on ONE GPU with a
world of one rank (RCCL all-to-all with itself), so everything but the wire is real - plan, gathers, local experts, combine.
Prints the step time next to the plain fused_experts call and a torch-profiler table of the GPU kernels."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sgl-cpu-tests_amd"))
import sgl_kernel  # noqa: E402,F401
from sgl_kernel.expert_parallel import ExpertParallelMoE  # noqa: E402

ops = torch.ops.sgl_kernel


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29733")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    M, K, N, E, topk = 16384, 2048, 768, 128, 8
    g = torch.Generator(device="cuda").manual_seed(1)
    w1 = ops.convert_weight_packed((torch.randn(E, 2 * N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
    w2 = ops.convert_weight_packed((torch.randn(E, K, N, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
    w1s = torch.randn(E, 2 * N // 128, K // 128, device="cuda", generator=g) * 1e-3
    w2s = torch.randn(E, K // 128, N // 128, device="cuda", generator=g) * 1e-3
    a = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    score = torch.softmax(torch.randn(M, E, device="cuda", generator=g).bfloat16(), dim=-1, dtype=torch.float32)
    tw, ids = torch.topk(score, topk)
    ids = ids.to(torch.int32)

    def local(h, w, lids):
        return ops.fused_experts_cpu(h, w1, w2, w, lids, False, False, True, w1s, w2s, [128, 128], None, None, True)

    ep = ExpertParallelMoE(E, local)

    def timed(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n

    t_plain = timed(lambda: local(a, tw, ids))
    t_ep = timed(lambda: ep(a, tw, ids))
    ref, out = local(a, tw, ids), ep(a, tw, ids)
    print(f"plain fused_experts {t_plain:.3f} ms, EP step (world 1) {t_ep:.3f} ms (+{(t_ep / t_plain - 1) * 100:.1f} %), "
          f"same bits: {torch.equal(ref, out)}", flush=True)
    for cf in (None, 1.0):
        epp = ExpertParallelMoE(E, local, capacity_factor=cf, profile=True)
        t = timed(lambda: epp(a, tw, ids))
        print(f"  split mode {'exact counts' if cf is None else 'capacity 1.0'}: {t:.3f} ms (+{(t / t_plain - 1) * 100:.1f} %), phases "
              f"{epp.phase_ms()}, same bits: {torch.equal(ref, epp(a, tw, ids))}, {epp.last_stats}", flush=True)
    # two steps in flight on two streams (what bench.py does for N > 1): step i+1's plan, gathers and dispatch overlap
    # step i's experts; every step owns its stream's workspace
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [None, None]

    def piped(n):
        for i in range(n):
            st = streams[i & 1]
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs[i & 1] = ep(a, tw, ids)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)

    piped(4)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    piped(10)
    e.record()
    torch.cuda.synchronize()
    print(f"EP steps alternating on two streams: {s.elapsed_time(e) / 10:.3f} ms per step, same bits: "
          f"{torch.equal(ref, outs[0])} {torch.equal(ref, outs[1])}", flush=True)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(3):
            ep(a, tw, ids)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
