"""world_size-2 gloo test (CPU) of the expert-parallel exchange logic (sgl_kernel/expert_parallel.py).

The communication plan, the id rewriting to local numbering with -1 padding
(/root/reference/test_moe_offloading_cpu.py:62-68) and the fixed-order combine are device-agnostic; here the
local experts are played by the oracle (test infrastructure) so the whole exchange can be checked without a GPU."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    for p in (ROOT, os.path.join(ROOT, "sgl-cpu-tests_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import recipes
        from oracle import moe
        from sgl_kernel.expert_parallel import ExpertParallelMoE, masked_allgather_reference
        M, N, K, E, topk, bn, bk = 37, 128, 128, 8, 3, 128, 128
        full = recipes.moe_fp8_inputs(M * world, N, K, E, topk, bn, bk, False, 5150)   # identical on every rank
        full["topk_ids"][3, 1] = -1                                                     # a padded slot survives EP
        sl = slice(rank * M, (rank + 1) * M)
        a, tw, ids = full["a"][sl], full["topk_weight"][sl], full["topk_ids"][sl]
        epr = E // world
        lo = rank * epr
        w1, w2 = full["w1"][lo:lo + epr], full["w2"][lo:lo + epr]
        w1s, w2s = full["w1s"][lo:lo + epr], full["w2s"][lo:lo + epr]

        def local(h, w, lids):
            return moe.fused_experts_fp8(h, w1, w2, w1s, w2s, (bn, bk), w, lids).bfloat16()

        ep = ExpertParallelMoE(E, local)
        out = ep(a, tw, ids)
        ref_full = moe.fused_experts_fp8(full["a"], full["w1"], full["w2"], full["w1s"], full["w2s"], (bn, bk),
                                         full["topk_weight"], full["topk_ids"])[sl]
        alt = masked_allgather_reference(a, tw, ids, E, local)
        ok1 = torch.allclose(ref_full.bfloat16(), out, rtol=1e-2, atol=1e-2)
        ok2 = torch.allclose(alt.float(), out.float(), rtol=2e-2, atol=1e-3)
        # every token row is sent at most once per destination rank
        tok, rk, cnt = ep.plan(ids)
        ok3 = len(set(zip(tok.tolist(), rk.tolist()))) == tok.numel() and int(cnt.sum()) == tok.numel()
        ret[rank] = (bool(ok1), bool(ok2), bool(ok3), ep.last_stats["rows_sent"])
    finally:
        dist.destroy_process_group()


def test_ep_all_to_all_matches_single_process_oracle():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        for r in range(world):
            ok1, ok2, ok3, sent = ret[r]
            assert ok1, f"rank {r}: EP result != full-expert oracle"
            assert ok2, f"rank {r}: all-to-all path != masked all-gather path"
            assert ok3, f"rank {r}: dispatch plan sends duplicate (token, rank) rows"
            assert sent > 0
