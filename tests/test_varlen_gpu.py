"""GPU parity of torch.ops.sgl_kernel.flash_attn_varlen_func against the golden outputs of the reference's own oracle
(flash_attn_varlen_ref, /root/reference/test_flash_attn_varlen.py:14-46) on the reference's cases (:110-115) plus one
longer causal case at the bench's head dim (:162).  Tolerance: the reference's utils.compare on bf16 (rtol = atol = 1e-2)
and a relative RMS error < 5e-3 against the fp32 oracle."""
import pytest
import torch

import recipes
from conftest import load_golden
from oracle import attention as oattn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import sgl_kernel  # noqa: F401
    assert torch.cuda.is_available()
    return torch.ops.sgl_kernel


@pytest.mark.parametrize("case", recipes.VARLEN_CASES, ids=lambda c: c[0])
def test_flash_attn_varlen(ops, case):
    name, batch, max_q, max_k, H, Hkv, D, DV, causal, varlen, seed = case
    g, _ = load_golden("varlen_" + name)
    inp = recipes.varlen_inputs(batch, max_q, max_k, H, Hkv, D, DV, varlen, seed)
    out = ops.flash_attn_varlen_func(inp["q"].cuda(), inp["k"].cuda(), inp["v"].cuda(), inp["cu_q"].cuda(), inp["cu_k"].cuda(),
                                     inp["max_q"], inp["max_k"], causal)
    assert out.shape == (inp["q"].shape[0], H, DV) and out.dtype == torch.bfloat16
    assert torch.allclose(g["ref_out"], out.cpu(), rtol=1e-2, atol=1e-2), name
    ref32 = oattn.flash_attn_varlen(inp["q"], inp["k"], inp["v"], inp["cu_q"], inp["cu_k"], causal)
    err = (out.float().cpu() - ref32).norm() / ref32.norm()
    assert err < 5e-3, f"{name}: relative RMS error {err:.2e}"


def test_flash_attn_varlen_strided_views_and_int64_offsets(ops):
    """q / k / v as head-slices of a packed qkv buffer (row stride != heads * dim), int64 cu_seqlens, an empty sequence."""
    g = torch.Generator().manual_seed(9301)
    H, D = 4, 64
    sq = torch.tensor([37, 0, 130, 1])
    sk = torch.tensor([64, 5, 129, 1])
    cu_q = torch.cat([torch.zeros(1, dtype=torch.int64), sq.cumsum(0)])
    cu_k = torch.cat([torch.zeros(1, dtype=torch.int64), sk.cumsum(0)])
    qkv_q = torch.randn(int(sq.sum()), 3 * H, D, generator=g).bfloat16()
    qkv_k = torch.randn(int(sk.sum()), 3 * H, D, generator=g).bfloat16()
    q, k, v = qkv_q[:, :H], qkv_k[:, H:2 * H], qkv_k[:, 2 * H:]
    for causal in (False, True):
        dq, dk = qkv_q.cuda(), qkv_k.cuda()
        out = ops.flash_attn_varlen_func(dq[:, :H], dk[:, H:2 * H], dk[:, 2 * H:], cu_q.cuda(), cu_k.cuda(), 130, 129, causal)
        ref = oattn.flash_attn_varlen(q, k, v, cu_q, cu_k, causal)
        assert torch.allclose(ref.bfloat16(), out.cpu(), rtol=1e-2, atol=1e-2)


def test_flash_attn_varlen_rejects_bad_arguments(ops):
    q = torch.zeros(8, 2, 64, dtype=torch.bfloat16, device="cuda")
    cu = torch.tensor([0, 8], dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError):
        ops.flash_attn_varlen_func(q.float(), q, q, cu, cu, 8, 8, False)
    with pytest.raises(RuntimeError):
        ops.flash_attn_varlen_func(q, q[:, :1].expand(8, 3, 64).contiguous(), q, cu, cu, 8, 8, False)
    with pytest.raises(RuntimeError):   # head dim beyond the built kernels
        big = torch.zeros(8, 2, 256, dtype=torch.bfloat16, device="cuda")
        ops.flash_attn_varlen_func(big, big, big, cu, cu, 8, 8, False)
