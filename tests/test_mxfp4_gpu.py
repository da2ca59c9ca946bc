"""GPU parity of mxfp4_scaled_mm_cpu / convert_scale_packed (/root/reference/test_mxfp4.py:146-210) against golden outputs
of the reference's own MXFP4QuantizeUtil + matmul expectation.  Pass predicate: utils.compare on bf16 (rtol = atol = 1e-2);
the weight expansion itself is exact, so the kernel is also compared with the fp32 oracle at relative RMS < 4e-3 (one bf16
rounding of the output)."""
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


@pytest.mark.parametrize("case", recipes.MXFP4_CASES, ids=lambda c: c[0])
def test_mxfp4_scaled_mm(ops, case):
    name, M, N, K, kind, has_bias, seed = case
    g, _ = load_golden("mxfp4_" + name)
    inp = recipes.mxfp4_inputs(M, N, K, kind, has_bias, seed, ogemm.mxfp4_quantize)
    assert torch.equal(inp["wq"], g["wq"]) and torch.equal(inp["ws"], g["ws"])
    a, wq, ws = inp["a"].cuda(), inp["wq"].cuda(), inp["ws"].cuda()
    bias = inp["bias"].cuda() if has_bias else None
    wp, sp = ops.convert_weight_packed(wq), ops.convert_scale_packed(ws)
    assert torch.equal(sp.cpu().flatten(), ogemm.scale_packed_order(inp["ws"]).flatten())   # test_mxfp4.py:186
    out = ops.mxfp4_scaled_mm_cpu(a, wp, sp, bias, True)
    assert out.shape == (M, N) and out.dtype == torch.bfloat16
    assert torch.allclose(g["ref_out"], out.cpu(), rtol=1e-2, atol=1e-2), name
    ref32 = ogemm.mxfp4_scaled_mm(inp["a"], inp["wq"], inp["ws"], inp["bias"])
    err = (out.float().cpu() - ref32).norm() / ref32.norm()
    assert err < 4e-3, f"{name}: relative RMS error {err:.2e}"
    # row-major scales (is_vnni=False) are the same computation
    assert torch.equal(out, ops.mxfp4_scaled_mm_cpu(a, wq, ws, bias, False))


def test_mxfp4_expansion_is_exact(ops):
    """x = identity: the output IS the dequantised weight matrix, bit for bit, for every nibble and a spread of scales."""
    N, K = 64, 256
    g = torch.Generator().manual_seed(9710)
    wq = torch.randint(0, 256, (N, K // 2), generator=g, dtype=torch.uint8)
    ws = torch.randint(100, 140, (N, K // 32), generator=g, dtype=torch.uint8)
    eye = torch.eye(K, dtype=torch.bfloat16, device="cuda")
    out = ops.mxfp4_scaled_mm_cpu(eye, wq.cuda(), ws.cuda(), None, False)
    assert torch.equal(out.float().cpu(), ogemm.mxfp4_dequant(wq, ws).t())


@pytest.mark.parametrize("M,N,K,kind,has_bias", [(64, 128, 256, "wide", False), (128, 512, 2048, "raw", True), (1000, 384, 1024, "wide", True),
                                                 (200, 256, 768, "quant", False)])
def test_mxfp4_native_fp4_mfma_path(ops, knob, M, N, K, kind, has_bias):
    """M >= 64, N % 128 == 0, K % 256 == 0 run on v_mfma_scale_f32_32x32x64_f8f6f4 with the E2M1 weights and E8M0 scales as
    stored (csrc/gemm_mxfp4.hip; activations as two e4m3 terms).  Same contract as the expansion path: the reference predicate
    against the fp32 oracle (/root/reference/test_mxfp4.py:166-170), relative RMS < 4e-3, both scale layouts identical bits, and
    close to the bf16-expansion path (SGLK_MXFP4_NATIVE=0) -- both carry exact products and differ only in summation order."""
    inp = recipes.mxfp4_inputs(M, N, K, kind, has_bias, 9800 + M, ogemm.mxfp4_quantize)
    a, wq, ws = inp["a"].cuda(), inp["wq"].cuda(), inp["ws"].cuda()
    bias = inp["bias"].cuda() if has_bias else None
    knob(SGLK_MXFP4_NATIVE=1)            # these shapes have too few tiles to be routed there by default
    out = ops.mxfp4_scaled_mm_cpu(a, wq, ws, bias, False)
    ref32 = ogemm.mxfp4_scaled_mm(inp["a"], inp["wq"], inp["ws"], inp["bias"])
    assert torch.allclose(ref32.bfloat16(), out.cpu(), rtol=1e-2, atol=1e-2)
    err = (out.float().cpu() - ref32).norm() / ref32.norm()
    assert err < 4e-3, f"relative RMS error {err:.2e}"
    assert torch.equal(out, ops.mxfp4_scaled_mm_cpu(a, ops.convert_weight_packed(wq), ops.convert_scale_packed(ws), bias, True))
    knob(SGLK_MXFP4_NATIVE=0)
    exp = ops.mxfp4_scaled_mm_cpu(a, wq, ws, bias, False)
    d = (out.float() - exp.float()).norm() / exp.float().norm()
    assert d < 3e-3, f"native vs expansion path {d:.2e}"


def test_mxfp4_native_path_is_exact_on_identity(ops, knob):
    """x = identity through the native path (256 rows): every output is ONE exact product -> the dequantised weights bit for bit,
    for every nibble, a spread of scales, and both lane halves / chunks of the operand layout."""
    N, K = 256, 512
    g = torch.Generator().manual_seed(9711)
    wq = torch.randint(0, 256, (N, K // 2), generator=g, dtype=torch.uint8)
    ws = torch.randint(100, 140, (N, K // 32), generator=g, dtype=torch.uint8)
    eye = torch.eye(K, dtype=torch.bfloat16, device="cuda")
    knob(SGLK_MXFP4_NATIVE=1)
    out = ops.mxfp4_scaled_mm_cpu(eye, wq.cuda(), ws.cuda(), None, False)
    assert torch.equal(out.float().cpu(), ogemm.mxfp4_dequant(wq, ws).t())
    # and with activations that are not powers of two: x = 1.375 * permutation (hi + lo both in use), scaled rows
    perm = torch.randperm(K, generator=g)
    x = torch.zeros(K, K)
    x[torch.arange(K), perm] = 1.375
    out = ops.mxfp4_scaled_mm_cpu(x.bfloat16().cuda(), wq.cuda(), ws.cuda(), None, False)
    want = (1.375 * ogemm.mxfp4_dequant(wq, ws).t()[perm]).bfloat16()
    assert torch.equal(out.cpu(), want)


def test_mxfp4_rejects_bad_arguments(ops):
    x = torch.zeros(2, 64, dtype=torch.bfloat16, device="cuda")
    wq = torch.zeros(32, 32, dtype=torch.uint8, device="cuda")
    ws = torch.zeros(32, 2, dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError):
        ops.mxfp4_scaled_mm_cpu(x, wq, ws[:, :1], None, True)
    with pytest.raises(RuntimeError):
        ops.mxfp4_scaled_mm_cpu(x.float(), wq, ws, None, True)
    with pytest.raises(RuntimeError):
        ops.mxfp4_scaled_mm_cpu(x[:, :48], wq[:, :24], ws, None, True)
