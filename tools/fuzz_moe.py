"""Randomised parity sweep of the fp8 (W8A16) fused_experts paths against the plain-C oracle, and (FUZZ_KIND=int8 | bf16) of the int8
W8A8 / bf16 paths against oracle/moe.py (test infrastructure, not a benchmark): random shapes, expert counts, routing (incl. -1 ids and skewed loads), block sizes, packed / row-major weights,
in-place or not.  Every kernel choice of the library (stream / mid / 128 / 256-row, generic engine) is hit by shape alone.
usage: python tools/fuzz_moe.py [iterations] [seed]      -> one line per failure, a summary line at the end"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sgl-cpu-tests_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sgl_kernel  # noqa: F401,E402
from sgl_kernel import _lib, _ops  # noqa: E402
import recipes  # noqa: E402
from oracle import c_oracle, moe  # noqa: E402

ops = torch.ops.sgl_kernel
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
KIND = os.environ.get("FUZZ_KIND", "fp8")
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
fails, paths = 0, {}
TIMES = os.environ.get("FUZZ_TIME", "0") == "1"
ONLY = int(os.environ.get("FUZZ_ONLY", "-1"))
slow = []


def routing(M, E, topk, kind, g):
    if kind == "softmax":
        return recipes.routing_softmax_topk(M, E, topk, g)
    if kind == "skewed":       # most tokens on two experts
        score = torch.randn(M, E, generator=g)
        score[:, : min(2, E)] += 6.0
        tw, ids = torch.topk(torch.softmax(score, -1), topk)
        return tw, ids.to(torch.int32)
    ids = torch.randint(0, E, (M, topk), generator=g, dtype=torch.int32)     # masked, repeats allowed
    ids[torch.rand(M, topk, generator=g) < 0.4] = -1
    return torch.randn(M, topk, generator=g), ids


for it in range(iters):
    N = rng.choice([128, 256, 384, 512, 768, 1024])
    K = rng.choice([128, 256, 512, 1024, 2048, 4096, 7168])      # up to 56 K blocks: the 128-token kernel's 64-block scale tables
    E = rng.choice([1, 2, 8, 16, 64])
    topk = rng.choice([t for t in (1, 2, 4, 8) if t <= E])
    M = rng.choice([1, 2, 3, 4, 7, 16, 33, 64, 100, 257, 600, 1500, 3000]) if E <= 16 else rng.choice([1, 4, 16, 64, 300, 1200])
    while M > 1 and 6.0 * M * topk * N * K > 2.5e10:      # keep the scalar oracle to a second or two per case
        M //= 2
    bn = rng.choice([64, 128])
    kind = rng.choice(["softmax", "softmax", "skewed", "masked"])
    packed = rng.random() < 0.8
    inplace = rng.random() < 0.5
    g = torch.Generator().manual_seed(rng.randrange(1 << 30))
    seed2 = rng.randrange(1 << 30) if KIND != "fp8" else 0
    if ONLY >= 0 and it != ONLY:      # FUZZ_ONLY=n: replay case n of this seed alone (same random draws up to here)
        continue
    if KIND != "fp8":
        # ---- int8 W8A8 (/root/reference/test_moe_int8.py:97-137: mean relative error < 1 %) and bf16 (test_moe.py:96-107) ----------
        E = min(E, 16)          # the torch oracle walks the experts
        topk = min(topk, E)
        tw, ids = routing(M, E, topk, kind, g)
        if KIND == "int8":
            inp = recipes.moe_int8_inputs(M, N, K, E, topk, seed2)
            ref = moe.fused_experts_int8(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], tw, ids).float()
            scales = (inp["w1s"].cuda(), inp["w2s"].cuda())
        else:
            inp = recipes.moe_bf16_inputs(M, N, K, E, topk, seed2)
            ref = moe.fused_experts_f32(inp["a"], inp["w1"].float(), inp["w2"].float(), tw, ids)
            scales = (None, None)
        k = 2.0 / max(float(ref.abs().max()), 1e-6)
        tw, ref = tw * k, ref * k
        w1d, w2d = inp["w1"].cuda(), inp["w2"].cuda()
        if packed:
            w1d, w2d = ops.convert_weight_packed(w1d), ops.convert_weight_packed(w2d)
        out = ops.fused_experts_cpu(inp["a"].cuda(), w1d, w2d, tw.cuda(), ids.cuda(), inplace, KIND == "int8", False, scales[0], scales[1],
                                    None, None, None, packed)
        torch.cuda.synchronize()
        o = out.float().cpu()
        ok_pred = torch.allclose(ref.bfloat16(), out.cpu(), rtol=1e-2, atol=1e-2)
        rel = float((o - ref).norm() / ref.norm().clamp_min(1e-12))
        mre = float((o - ref).abs().mean() / ref.abs().mean().clamp_min(1e-12))
        path = _ops.last_path & _lib.PATH_TILE_MASK
        paths[path] = paths.get(path, 0) + 1
        # int8: the reference's own predicate (test_moe_int8.py:134-137, mean relative error < 1 %).  The elementwise allclose also
        # held in every case but one of 600 -- seed 778, case 209: one token routed to the same expert twice with weights -3.58 and
        # +3.40, so the two bf16-rounded rows cancel and leave 0.02 of rounding on elements near zero (both weight layouts give the
        # same bits; replay with FUZZ_ONLY=209) -- so it is required only where no routed expert repeats in a token.
        repeats = any(len(set(e for e in row if e >= 0)) < sum(1 for e in row if e >= 0) for row in ids.tolist())
        bad = (mre > 0.01 or (not ok_pred and not repeats)) if KIND == "int8" else (not ok_pred or rel > 6e-3)
        if ONLY >= 0:
            err = (o - ref).abs()
            tol = 1e-2 + 1e-2 * ref.bfloat16().float().abs()
            idx = torch.nonzero(err > tol)
            print(f"case {it}: ids={ids.tolist()} tw={[round(float(x), 3) for x in tw.flatten()]} max|ref|={float(ref.abs().max()):.3f} "
                  f"elements over tolerance: {idx.shape[0]} of {o.numel()}, worst err {float(err.max()):.4f} at ref {float(ref.flatten()[err.argmax()]):.4f}")
            o2 = ops.fused_experts_cpu(inp["a"].cuda(), inp["w1"].cuda(), inp["w2"].cuda(), tw.cuda(), ids.cuda(), False, KIND == "int8", False,
                                       scales[0], scales[1], None, None, None, False).float().cpu()
            print("  unpacked-weights call (another kernel): same bits" if torch.equal(o2, o) else f"  unpacked-weights call differs: max {float((o2 - o).abs().max()):.4f}, its worst err {float((o2 - ref).abs().max()):.4f}")
        if bad or not torch.isfinite(o).all():
            fails += 1
            print(f"FAIL {KIND} it={it} M={M} N={N} K={K} E={E} topk={topk} {kind} packed={packed} inplace={inplace} path={_ops.last_path:#x} "
                  f"rel={rel:.2e} mre={mre:.2e} pred={ok_pred}", flush=True)
        continue
    a = (torch.randn(M, K, generator=g) / K ** 0.5).bfloat16()
    w1, w2 = recipes.fp8_weight((E, 2 * N, K), g), recipes.fp8_weight((E, K, N), g)
    w1s = torch.randn(E, 2 * N // bn, K // 128, generator=g) * recipes.SCALE_FACTOR
    w2s = torch.randn(E, K // bn, N // 128, generator=g) * recipes.SCALE_FACTOR
    tw, ids = routing(M, E, topk, kind, g)
    ref = c_oracle.fused_experts_fp8(a, w1, w2, w1s, w2s, (bn, 128), tw, ids)
    # outputs into the O(1) range the reference's atol = rtol = 1e-2 is meant for (its tests give |out| <~ 2; at |out| in [2, 4)
    # two bf16 ulps already exceed it): the result is linear in the routing weights
    k = 2.0 / max(float(ref.abs().max()), 1e-6)
    tw, ref = tw * k, ref * k
    d = [t.cuda() for t in (a, w1, w2, tw, ids, w1s, w2s)]
    w1d, w2d = (ops.convert_weight_packed(d[1]), ops.convert_weight_packed(d[2])) if packed else (d[1], d[2])
    out = ops.fused_experts_cpu(d[0], w1d, w2d, d[3], d[4], inplace, False, True, d[5], d[6], [bn, 128], None, None, packed)
    torch.cuda.synchronize()
    o = out.float().cpu()
    ok_pred = torch.allclose(ref.bfloat16(), out.cpu(), rtol=1e-2, atol=1e-2)
    rel = float((o - ref).norm() / ref.norm().clamp_min(1e-12))
    path = _ops.last_path & _lib.PATH_TILE_MASK
    paths[path] = paths.get(path, 0) + 1
    if TIMES:      # FUZZ_TIME=1: time the case, keep its distance from max(flops / 2.5 PF, touched weight bytes / 8 TB/s)
        fn = lambda: ops.fused_experts_cpu(d[0], w1d, w2d, d[3], d[4], False, False, True, d[5], d[6], [bn, 128], None, None, packed)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        valid = ids[(ids >= 0) & (ids < E)]
        touched = int(torch.unique(valid).numel())
        t_min = max(6.0 * valid.numel() * N * K / 2.5e15, touched * 3.0 * N * K / 8e12) * 1e3
        if t_min > 3e-3:
            slow.append((t_min / ms, ms, f"M={M} N={N} K={K} E={E} topk={topk} bn={bn} {kind} packed={packed} path={path} rows/expert={valid.numel() / max(touched, 1):.0f}"))
    if not ok_pred or rel > 6e-3 or not torch.isfinite(o).all():
        fails += 1
        diff = (o - ref.bfloat16().float()).abs()
        bad = diff > 1e-2 + 1e-2 * o.abs()
        print(f"FAIL it={it} M={M} N={N} K={K} E={E} topk={topk} bn={bn} {kind} packed={packed} inplace={inplace} "
              f"path={_ops.last_path:#x} rel={rel:.2e} pred={ok_pred} bad={int(bad.sum())}/{bad.numel()} max|diff|={float(diff.max()):.4f} "
              f"max|ref|={float(ref.abs().max()):.2f} max|tw|={float(tw.abs().max()):.2f}", flush=True)
for frac, ms, desc in sorted(slow)[:20]:
    print(f"  slow: {frac:.3f} of its roofline, {ms:.4f} ms  {desc}")
print(f"fuzz_moe {KIND}: {iters} cases, {fails} failures, tile paths {dict(sorted(paths.items()))}")
sys.exit(1 if fails else 0)
