#!/usr/bin/env python3
"""Randomised parity sweep of grouped_topk_cpu (softmax variant) against oracle/routing.py: ids BIT-EXACT, weights within 2e-5.
Tie-heavy on purpose: logits drawn from a few levels (zeros of both signs among them), all three gating dtypes, expert counts that
are not powers of two, every group / top-k split, selections that run out of selected-group experts (the second-chance picks).

usage: python tools/fuzz_topk.py [iterations] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sgl-cpu-tests_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import sgl_kernel  # noqa: E402,F401
from oracle import routing  # noqa: E402

ops = torch.ops.sgl_kernel
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20261005)
fails = 0
for it in range(iters):
    E = rng.choice([4, 8, 16, 24, 32, 64, 96, 128, 160, 256, 384, 512, 1024])
    G = rng.choice([g for g in (1, 2, 3, 4, 8, 16, 32, 64) if E % g == 0 and g <= E])
    topk_group = rng.randint(1, G)
    topk = rng.randint(1, min(E, 64, 12))
    M = rng.choice([1, 2, 5, 16, 17, 64, 300])
    dt = rng.choice([torch.float32, torch.bfloat16, torch.float16])
    g = torch.Generator().manual_seed(rng.randrange(1 << 30))
    style = rng.choice(["levels", "levels", "normal", "flat"])
    if style == "levels":          # a handful of values, +0 / -0 among them: ties everywhere
        levels = torch.tensor([-2.5, -1.0, -0.0, 0.0, 0.5, 0.5, 3.0])
        gating = levels[torch.randint(0, len(levels), (M, E), generator=g)]
    elif style == "flat":
        gating = torch.zeros(M, E)
    else:
        gating = torch.randn(M, E, generator=g) * 3
    gating = gating.to(dt)
    renorm = rng.random() < 0.5
    ow, oids = routing.grouped_topk(gating, topk, renorm, G, topk_group)
    w, ids = ops.grouped_topk_cpu(gating.cuda(), gating.cuda(), topk, renorm, G, topk_group, 0, None, None)
    ids, w = ids.cpu(), w.cpu()
    same = torch.equal(ids.to(torch.int32), oids)
    # a renormalised row whose picks all weigh 0 is 0 / 0 on both sides
    wok = torch.allclose(torch.nan_to_num(w), torch.nan_to_num(ow), rtol=2e-5, atol=1e-6) and torch.equal(torch.isnan(w), torch.isnan(ow))
    if not same or not wok:
        fails += 1
        bad = torch.nonzero((ids.to(torch.int32) != oids).any(dim=1)).flatten().tolist()[:3]
        print(f"FAIL it={it} M={M} E={E} G={G} topk_group={topk_group} topk={topk} {dt} {style} renorm={renorm} ids_equal={same} weights_ok={wok} "
              f"rows {bad}: kernel {[ids[r].tolist() for r in bad]} oracle {[oids[r].tolist() for r in bad]}", flush=True)
print(f"fuzz_topk: {iters} cases, {fails} failures")
