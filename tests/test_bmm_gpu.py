"""GPU parity of bmm_cpu (/root/reference/test_bmm_fp8.py:38-39,52-74) against golden torch.bmm outputs on the reference's
shapes and views: row-major mat2 (is_vnni=False) and convert_weight_packed(mat2) (is_vnni=True), out / mat1 as transposed,
narrowed views.  Pass predicate: utils.compare on bf16 (rtol = atol = 1e-2, /root/reference/utils.py:3-13)."""
import pytest
import torch

import recipes
from conftest import load_golden
from oracle import gemm as ogemm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import sgl_kernel  # noqa: F401
    assert torch.cuda.is_available()
    return torch.ops.sgl_kernel


def device_views(B, M, N, K, chunk, inp):
    """Rebuild the reference's views on the device (a .cuda() of a view would come back contiguous)."""
    pad = 64 if chunk else 0
    a = torch.zeros(M, B, K + pad, dtype=torch.bfloat16, device="cuda").narrow(2, 0, K).transpose(0, 1)
    a.copy_(inp["mat1"])
    out = torch.full((M, B, N + pad), float("nan"), dtype=torch.bfloat16, device="cuda").narrow(2, 0, N).transpose(0, 1)
    return a, out


@pytest.mark.parametrize("case", recipes.BMM_CASES, ids=lambda c: c[0])
def test_bmm(ops, case):
    name, B, M, N, K, chunk, seed = case
    g, _ = load_golden("bmm_" + name)
    inp = recipes.bmm_inputs(B, M, N, K, chunk, seed)
    a, out = device_views(B, M, N, K, chunk, inp)
    w = inp["mat2"].cuda()
    assert ops.bmm_cpu(out, a, w, False, None) is None
    assert torch.allclose(g["ref_out"], out.cpu(), rtol=1e-2, atol=1e-2), name
    ref32 = ogemm.bmm(inp["mat1"], inp["mat2"])
    assert torch.allclose(ref32.bfloat16(), out.cpu(), rtol=1e-2, atol=1e-2)
    first = out.clone()
    out.fill_(float("nan"))
    ops.bmm_cpu(out, a, ops.convert_weight_packed(w), True, None)
    assert torch.equal(first, out), "packed and row-major mat2 must give the same bits (same fp32 summation order)"


def test_bmm_rejects_bad_arguments(ops):
    a = torch.zeros(2, 3, 64, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(2, 32, 64, dtype=torch.bfloat16, device="cuda")
    out = torch.zeros(2, 3, 32, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError):
        ops.bmm_cpu(out, a, w[:, :, :32], False, None)
    with pytest.raises(RuntimeError):
        ops.bmm_cpu(out, a, w, False, torch.ones(1, device="cuda"))
    with pytest.raises(RuntimeError):
        ops.bmm_cpu(out.float(), a, w, False, None)
