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
    pairs = member.t().nonzero()                       # sorted by rank, then token: the order `back` arrives in
    send_rank, send_tok = pairs[:, 0], pairs[:, 1]
    send_l = member.sum(dim=0).tolist()
    back = torch.randn(pairs.shape[0], K, generator=g).bfloat16()
    ep = object.__new__(ExpertParallelMoE)
    ep.world = G
    host = ep._reduce(back, send_tok, send_rank, send_l, M)
    dev = ep._reduce(back.cuda(), send_tok.cuda(), send_rank.cuda(), send_l, M)
    assert dev.dtype == torch.bfloat16 and dev.is_cuda
    assert torch.equal(host, dev.cpu())
    assert torch.count_nonzero(dev[5]) == 0
