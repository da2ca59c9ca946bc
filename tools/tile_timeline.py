#!/usr/bin/env python3
"""Developer tool (needs a SGLK_DEV_ABLATE build: `SGLK_DEV_ABLATE=1 python sgl-cpu-tests_amd/build.py`, then run this with
SGLK_LIB_PATH=sgl-cpu-tests_amd/sgl_kernel/libsglk_dev.so): per-workgroup timeline of the two grouped GEMMs of fused_experts
at the bench shape -- prologue / main loop / epilogue durations and the gap between consecutive workgroups on a CU.
SGLK_FP8_ACT=1 in the environment looks at the a8 kernels instead."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sgl-cpu-tests_amd"))
import torch
import sgl_kernel  # noqa
ops = torch.ops.sgl_kernel
K, N, E, topk, M = 2048, 768, 128, 8, int(os.environ.get("SGLK_TL_M", "16384"))
g = torch.Generator(device="cuda").manual_seed(1)
w1 = ops.convert_weight_packed((torch.randn(E, 2 * N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
w2 = ops.convert_weight_packed((torch.randn(E, K, N, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn))
w1s = torch.randn(E, 2 * N // 128, K // 128, device="cuda", generator=g) * 1e-3
w2s = torch.randn(E, K // 128, N // 128, device="cuda", generator=g) * 1e-3
a = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
tw, ids = torch.topk(torch.softmax(torch.randn(M, E, device="cuda", generator=g), dim=-1), topk); ids = ids.to(torch.int32)
dbg = torch.zeros(2 * 32 * 16384, dtype=torch.int64, device="cuda")
os.environ["SGLK_DBG_PTR"] = hex(dbg.data_ptr())
f = lambda: ops.fused_experts_cpu(a, w1, w2, tw, ids, False, False, True, w1s, w2s, [128, 128], None, None, True)
if os.environ.get("SGLK_TL_INT8"):   # the int8 W8A8 operator on the same kernels
    w1 = ops.convert_weight_packed(torch.randint(-127, 128, (E, 2 * N, K), device="cuda", generator=g, dtype=torch.int8))
    w2 = ops.convert_weight_packed(torch.randint(-127, 128, (E, K, N), device="cuda", generator=g, dtype=torch.int8))
    w1s = torch.rand(E, 2 * N, device="cuda", generator=g) * 1e-2
    w2s = torch.rand(E, K, device="cuda", generator=g) * 1e-2
    f = lambda: ops.fused_experts_cpu(a, w1, w2, tw, ids, False, True, False, w1s, w2s, None, None, None, True)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(50): f()
    torch.cuda.synchronize()
dbg.zero_(); torch.cuda.synchronize()
f(); torch.cuda.synchronize()
both = dbg.cpu().view(2, -1, 32)
for name, full in (("GEMM-1", both[0]), ("GEMM-2", both[1])):
    full = full[full[:, 19] > 0].double()
    entry, l0, l1, st, ack = full[:, 18], full[:, 19], full[:, 20], full[:, 21], full[:, 24]
    us = lambda x: x / 100.0
    print(f"{name}: {len(full)} workgroups; kernel span {us(ack.max() - entry.min()):.1f} us")
    e1, e2, e3 = full[:, 25], full[:, 26], full[:, 27]
    if (full[:, 28] > 0).any():   # int8 GEMM-1: the row-maximum exchange between the m-tile's workgroups
        x1, x2, x3 = full[:, 28], full[:, 29], full[:, 30]
        for lab, v in (("  epi: barrier 1 -> maxima posted", x1 - e1), ("  epi: arrival + wait", x2 - x1), ("  epi: maxima read back", x3 - x2),
                       ("  epi: quantise -> image written", e2 - x3)):
            print(f"   {lab:40s} median {v.median() / 100.0:7.2f} us   p10 {v.quantile(0.1) / 100.0:7.2f}   p90 {v.quantile(0.9) / 100.0:7.2f}")
    for lab, v in (("  epi: loop end -> barrier 1", e1 - l1), ("  epi: image written by wave 0", e2 - e1), ("  epi: barrier 2", e3 - e2),
                   ("  epi: rows read + stores issued", st - e3), ("prologue (entry -> loop)", l0 - entry), ("main loop", l1 - l0), ("epilogue (loop end -> stores issued)", st - l1),
                   ("store ack", ack - st), ("whole workgroup", ack - entry)):
        print(f"   {lab:40s} median {us(v.median()):7.2f} us   p10 {us(v.quantile(0.1)):7.2f}   p90 {us(v.quantile(0.9)):7.2f}")
    cu = (full[:, 23].long() << 32) | (full[:, 22].long() & 0xFF00)   # xcc | se/sh/cu bits of HW_ID
    gaps, per_cu = [], []
    for c in cu.unique():
        sel = full[cu == c]
        order = sel[:, 18].argsort()
        sel = sel[order]
        per_cu.append(len(sel))
        if len(sel) > 1:
            gaps.append(sel[1:, 18] - sel[:-1, 24])
    gaps = torch.cat(gaps)
    # co-resident workgroups (two per CU on the 128-token kernels): share of a CU's busy span in which at least one / more than
    # one workgroup is inside its main loop (the matrix pipe has work / is shared)
    cov1, cov2 = [], []
    for c in cu.unique():
        sel = full[cu == c]
        ev = sorted([(float(x), 1) for x in sel[:, 19]] + [(float(x), -1) for x in sel[:, 20]])
        span = float(sel[:, 24].max() - sel[:, 18].min())
        depth, last, t1, t2 = 0, ev[0][0], 0.0, 0.0
        for t, d in ev:
            if depth >= 1: t1 += t - last
            if depth >= 2: t2 += t - last
            depth += d
            last = t
        cov1.append(t1 / span)
        cov2.append(t2 / span)
    cov1, cov2 = torch.tensor(cov1), torch.tensor(cov2)
    # rate of a workgroup's main loop alone on its CU (r1) and beside a co-resident main loop (r2, per workgroup): least squares
    # of  1 = d_alone * r1 + d_shared * r2  over the workgroups (units: main loops per us)
    rows_a, rows_b = [], []
    for c in cu.unique():
        sel = full[cu == c]
        for i in range(len(sel)):
            a0, a1 = float(sel[i, 19]), float(sel[i, 20])
            sh = 0.0
            for j in range(len(sel)):
                if j != i:
                    sh += max(0.0, min(a1, float(sel[j, 20])) - max(a0, float(sel[j, 19])))
            rows_a.append((a1 - a0 - sh) / 100.0)
            rows_b.append(sh / 100.0)
    A = torch.tensor([rows_a, rows_b], dtype=torch.float64).T
    sol = torch.linalg.lstsq(A, torch.ones(len(rows_a), 1, dtype=torch.float64)).solution.flatten()
    print(f"   main loop alone on the CU: {1 / sol[0]:.2f} us per tile; beside another main loop: {1 / sol[1]:.2f} us per tile "
          f"(= {0.5 / sol[1]:.2f} us of CU time per tile)")
    if (full[:, 0] > 0).any():
        clk = full[:, 0] / full[:, 1].clamp_min(1) * 100.0
        print(f"   in-kernel clock over the main loop: median {clk.median():.0f} MHz (p10 {clk.quantile(0.1):.0f}, p90 {clk.quantile(0.9):.0f})")
    print(f"   per CU: some workgroup in its main loop {cov1.median():.3f} of the busy span (p10 {cov1.quantile(0.1):.3f}); two or more {cov2.median():.3f}")
    # how the launch ends: when each CU goes idle relative to the last one, and the work (main-loop time) per CU
    t_end = ack.max()
    idle, busy = [], []
    for c in cu.unique():
        sel = full[cu == c]
        idle.append(float(t_end - sel[:, 24].max()) / 100.0)
        busy.append(float((sel[:, 20] - sel[:, 19]).sum()) / 100.0)
    idle, busy = torch.tensor(idle), torch.tensor(busy)
    nta = ((full[:, 20] - full[:, 19]) < 0.6 * (l1 - l0).median()).sum()
    print(f"   end of the launch: a CU is idle for the last {idle.median():.1f} us (median; p90 {idle.quantile(0.9):.1f}, max {idle.max():.1f}) of "
          f"{us(ack.max() - entry.min()):.1f}; main-loop time per CU min/median/max {busy.min():.0f}/{busy.median():.0f}/{busy.max():.0f} us; "
          f"{int(nta)} workgroups with a main loop under 0.6 of the median (partial tiles)")
    print(f"   CUs seen {len(per_cu)}, workgroups per CU min/max {min(per_cu)}/{max(per_cu)}; gap between consecutive workgroups on a CU: "
          f"median {us(gaps.median()):.2f} us  p10 {us(gaps.quantile(0.1)):.2f}  p90 {us(gaps.quantile(0.9)):.2f}")
