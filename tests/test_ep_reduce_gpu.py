"""The expert-parallel combine on the GPU (sglk_ep_reduce_rows through ExpertParallelMoE._reduce) against the torch
formulation the CPU / gloo path uses: same sums in the same (rank) order -> identical bits."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("G", [2, 8])
def test_ep_reduce_matches_host_formulation(G):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sgl-cpu-tests_amd"))
    import sgl_kernel  # noqa: F401
    from sgl_kernel.expert_parallel import ExpertParallelMoE
    assert torch.cuda.is_available()
    M, K = 777, 2048
    g = torch.Generator().manual_seed(31 + G)
    member = torch.rand(M, G, generator=g) < 0.6
    member[5] = False                                  # a token nobody returns anything for -> zeros
    # the plan's view of it: pos[m][d] = index of token m inside rank d's segment, segments back to back
    counts = member.sum(dim=0).to(torch.int32)
    pos = torch.where(member, torch.cumsum(member.to(torch.int32), dim=0) - 1, torch.full((M, G), -1, dtype=torch.int32)).to(torch.int32)
    seg_start = torch.zeros(G + 1, dtype=torch.int32)
    seg_start[1:] = torch.cumsum(counts, 0)
    back = torch.randn(int(seg_start[G]), K, generator=g).bfloat16()
    ep = object.__new__(ExpertParallelMoE)
    ep.world = G
    host = ep._reduce(back, pos, seg_start, M)
    dev = ep._reduce(back.cuda(), pos.cuda(), seg_start.cuda(), M)
    assert dev.dtype == torch.bfloat16 and dev.is_cuda
    assert torch.equal(host, dev.cpu())
    assert torch.count_nonzero(dev[5]) == 0


@pytest.mark.parametrize("M,topk,E,G,capacity", [(1, 8, 128, 8, 0), (777, 8, 128, 8, 0), (16384, 8, 128, 8, 0), (300, 3, 8, 4, 0),
                                                 (300, 3, 8, 4, 300), (300, 3, 8, 2, 120), (5000, 8, 128, 8, 5000)])
def test_ep_plan_and_pack_kernels_match_torch_formulation(M, topk, E, G, capacity):
    """sglk_ep_plan / sglk_ep_pack (one launch each) against the torch formulation the gloo tests exercise, bit for bit:
    counts, segment starts, positions (ascending token order), overflow mask, payload rows incl. the id rewrite with -1 and,
    in capacity mode, the -1 ids of unused rows."""
    import types

    from sgl_kernel import expert_parallel as epm
    g = torch.Generator().manual_seed(M * 31 + G)
    ids = torch.randint(0, E, (M, topk), generator=g, dtype=torch.int32)
    ids[torch.rand(M, topk, generator=g) < 0.1] = -1
    if M > 10:
        ids[7] = -1                                  # a token that goes nowhere
        ids[8] = E + 5                               # out-of-range ids are treated like -1
    K = 256
    hidden = torch.randn(M, K, generator=g).bfloat16()
    tw = torch.rand(M, topk, generator=g)
    row_bytes = (2 * K + 8 * topk + 15) // 16 * 16
    c_ref, s_ref, p_ref, ovf_ref = epm.plan_torch(ids, E, G, capacity)
    fake = types.SimpleNamespace(world=G, num_experts=E, last_overflow=None)
    counts, seg_start, pos = epm.ExpertParallelMoE.plan(fake, ids.cuda(), capacity)
    assert torch.equal(counts.cpu(), c_ref) and torch.equal(seg_start.cpu(), s_ref) and torch.equal(pos.cpu(), p_ref)
    assert int(fake.last_overflow.item()) == ovf_ref
    if capacity and capacity < M:
        assert ovf_ref != 0, "this case is meant to overflow"
    rows = int(s_ref[G])
    payload = epm.ExpertParallelMoE.pack(fake, hidden.cuda(), tw.cuda(), ids.cuda(), pos, seg_start, counts, rows, row_bytes, capacity)
    ref = epm.pack_torch(hidden, tw, ids, p_ref, s_ref, c_ref, row_bytes, E, G, capacity)
    got = payload.cpu()
    used = torch.zeros(rows, dtype=torch.bool)
    for d in range(G):
        n = min(int(c_ref[d]), capacity) if capacity else int(c_ref[d])
        used[int(s_ref[d]):int(s_ref[d]) + n] = True
    assert torch.equal(got[used][:, :2 * K + 8 * topk], ref[used][:, :2 * K + 8 * topk]), "payload rows differ"
    # unused rows of a capacity segment: only their ids are defined (-1)
    assert torch.equal(got[~used][:, 2 * K:2 * K + 4 * topk], ref[~used][:, 2 * K:2 * K + 4 * topk])
