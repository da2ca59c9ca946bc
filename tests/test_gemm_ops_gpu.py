"""GPU parity of the GEMM-shaped operators on the generic engine, through torch.ops.sgl_kernel -> C-ABI -> HIP:
bf16 / int8 fused_experts (/root/reference/test_moe.py, test_moe_int8.py), shared_expert (test_shared_experts.py,
test_moe_fp8_ext.py:27-67), weight_packed_linear / fp8_scaled_mm / int8_scaled_mm (test_gemm*.py).
Expected values: tests/golden (the reference's own oracles run in the build container)."""
import pytest
import torch

import recipes
from conftest import load_golden
from oracle import gemm as ogemm
from oracle import moe

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import sgl_kernel  # noqa: F401
    assert torch.cuda.is_available()
    return torch.ops.sgl_kernel


def cuda(d):
    return {k: v.cuda() for k, v in d.items()}


def ref_pred(ref, out):
    """utils.compare's predicate (/root/reference/utils.py:9-13) on bf16."""
    return torch.allclose(ref.bfloat16().cpu(), out.bfloat16().cpu(), rtol=1e-2, atol=1e-2)


def rel_rms(out, ref):
    return float((out.float().cpu() - ref.float().cpu()).norm() / ref.float().cpu().norm().clamp_min(1e-12))


# ---- fused_experts bf16 / int8 ------------------------------------------------------------------------------------
@pytest.mark.parametrize("prepack", [False, True])
@pytest.mark.parametrize("case", recipes.MOE_BF16_CASES, ids=lambda c: c[0])
def test_fused_experts_bf16(ops, case, prepack):
    from sglang.srt.layers.amx_utils import CPUQuantMethod
    name, M, N, K, E, topk, renorm, seed, full = case
    g, _ = load_golden("moe_bf16_" + name)
    inp = cuda(recipes.moe_bf16_inputs(M, N, K, E, topk, seed))
    w, ids = moe.softmax_topk(inp["score"].cpu(), topk, renorm)
    w1 = ops.convert_weight_packed(inp["w1"]) if prepack else inp["w1"]
    w2 = ops.convert_weight_packed(inp["w2"]) if prepack else inp["w2"]
    a = inp["a"].clone()
    # 13-argument CPUQuantMethod form, inplace=True, exactly as /root/reference/test_moe.py:79-92
    out = ops.fused_experts_cpu(a, w1, w2, w.cuda(), ids.cuda(), True, CPUQuantMethod.UNQUANT, None, None, None, None,
                                None, prepack)
    assert out.data_ptr() == a.data_ptr()
    assert ref_pred(g["ref_out"], out), name
    ref32 = moe.fused_experts_f32(inp["a"].cpu(), inp["w1"].cpu().float(), inp["w2"].cpu().float(), w, ids)
    assert rel_rms(out, ref32) < 6e-3


@pytest.mark.parametrize("prepack", [False, True])
@pytest.mark.parametrize("case", recipes.MOE_INT8_CASES, ids=lambda c: c[0])
def test_fused_experts_int8(ops, case, prepack):
    name, M, N, K, E, topk, seed, full = case
    g, _ = load_golden("moe_int8_" + name)
    inp = cuda(recipes.moe_int8_inputs(M, N, K, E, topk, seed))
    w1 = ops.convert_weight_packed(inp["w1"]) if prepack else inp["w1"]
    w2 = ops.convert_weight_packed(inp["w2"]) if prepack else inp["w2"]
    out = ops.fused_experts_cpu(inp["a"].clone(), w1, w2, inp["topk_weight"], inp["topk_ids"], True, True, False,
                                inp["w1s"], inp["w2s"], None, None, None, prepack)
    ref = g["ref_out"].float()
    # the reference's real bar for int8 (test_moe_int8.py:134-137): mean relative error < 1 %
    mre = (out.float().cpu() - ref).abs().mean() / ref.abs().mean()
    assert mre < 0.01, f"{name}: mean relative error {mre:.4f}"
    assert ref_pred(ref, out), name


@pytest.mark.parametrize("shape", [(1024, 768, 2048, 16, 4), (1000, 256, 512, 4, 2), (3000, 384, 1024, 32, 8)],
                         ids=lambda s: "x".join(map(str, s)))
def test_fused_experts_int8_on_int8_mfma(ops, shape):
    """Large-M packed int8 fused_experts runs both grouped GEMMs on mfma_i32_32x32x32_i8 (csrc/gemm_i8_256.hip); checked
    against the restated oracle (oracle/moe.py, /root/reference/test_moe_int8.py:59-94) with the reference's own bars,
    against the generic engine (row-major weights), for ragged expert loads and masked slots."""
    M, N, K, E, topk = shape
    inp = recipes.moe_int8_inputs(M, N, K, E, topk, 4000 + M)
    ids = inp["topk_ids"].clone()
    ids[::7, 0] = -1                                         # some masked slots (offloading contract)
    ref = moe.fused_experts_int8(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], inp["topk_weight"], ids).float()
    d = cuda(inp)
    idc = ids.cuda()
    w1p, w2p = ops.convert_weight_packed(d["w1"]), ops.convert_weight_packed(d["w2"])
    out = ops.fused_experts_cpu(d["a"].clone(), w1p, w2p, d["topk_weight"], idc, False, True, False, d["w1s"], d["w2s"],
                                None, None, None, True)
    mre = (out.float().cpu() - ref).abs().mean() / ref.abs().mean()
    assert mre < 0.01, f"mean relative error {mre:.4f}"
    assert ref_pred(ref, out)
    generic = ops.fused_experts_cpu(d["a"].clone(), d["w1"], d["w2"], d["topk_weight"], idc, False, True, False, d["w1s"],
                                    d["w2s"], None, None, None, False)
    assert rel_rms(out, generic) < 5e-3
    again = ops.fused_experts_cpu(d["a"].clone(), w1p, w2p, d["topk_weight"], idc, False, True, False, d["w1s"], d["w2s"],
                                  None, None, None, True)
    assert torch.equal(out, again), "run-to-run bit identity"


@pytest.mark.parametrize("shape", [(1024, 768, 2048, 16, 4), (1000, 256, 512, 4, 2), (3000, 384, 1024, 32, 8), (4096, 768, 2048, 128, 8),
                                   (16384, 768, 2048, 128, 8)], ids=lambda s: "x".join(map(str, s)))
def test_fused_experts_int8_on_the_128_token_kernel(ops, knob, shape):
    """Large-M int8 fused_experts on moe_gemm_fp8w_s128.hip (terms = 0, the default): exact int32 sums on mfma_i32_32x32x32_i8,
    silu(gate) * up quantised per token inside GEMM-1's epilogue (row maxima exchanged between the m-tile's workgroups).  Same
    arithmetic as gemm_i8_256.hip + the separate quantisation pass: BIT-IDENTICAL to that path (SGLK_I8_S128=0), ragged expert
    loads, masked slots and Qwen3-30B-A3B expert dims at 4096 / 16384 tokens (bench_moe.py:89-106) included; the small shapes
    also against the oracle with the reference's bars (test_moe_int8.py:134-137)."""
    from sgl_kernel import _lib, _ops
    M, N, K, E, topk = shape
    inp = recipes.moe_int8_inputs(M, N, K, E, topk, 4200 + M)
    ids = inp["topk_ids"].clone()
    ids[::7, 0] = -1
    d = cuda(inp)
    idc = ids.cuda()
    w1p, w2p = ops.convert_weight_packed(d["w1"]), ops.convert_weight_packed(d["w2"])
    call = lambda: ops.fused_experts_cpu(d["a"].clone(), w1p, w2p, d["topk_weight"], idc, False, True, False, d["w1s"], d["w2s"],
                                         None, None, None, True)
    knob(SGLK_I8_S128=1)
    out = call()
    assert (_ops.last_path & _lib.PATH_TILE_MASK) == 128
    assert torch.equal(out, call()), "run-to-run bit identity"
    knob(SGLK_I8_S128=0)
    out256 = call()
    assert (_ops.last_path & _lib.PATH_TILE_MASK) == 256
    assert torch.isfinite(out.float()).all()
    assert torch.equal(out, out256), "128-token int8 kernel != 256-row int8 kernels"
    if M <= 4096:
        sel = torch.arange(0, M, max(1, M // 256))
        ref = moe.fused_experts_int8(inp["a"][sel], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], inp["topk_weight"][sel], ids[sel]).float()
        o = out[sel].float().cpu()
        mre = (o - ref).abs().mean() / ref.abs().mean()
        assert mre < 0.01, f"mean relative error {mre:.4f}"
        assert ref_pred(ref, out[sel])


@pytest.mark.parametrize("shape", [(200, 768, 2048, 16, 4), (4, 384, 1024, 32, 8), (150, 384, 640, 8, 2), (61, 256, 4352, 4, 2),
                                   (300, 256, 512, 8, 2)], ids=lambda s: "x".join(map(str, s)))
def test_fused_experts_int8_on_mid_kernel(ops, shape):
    """Small / mid-size packed int8 fused_experts (below 44 rows per expert) runs on the weight-streaming kernel
    csrc/gemm_i8_mid.hip (mfma_i32_16x16x64_i8): even / odd / long K-block counts, tiles up to 128 rows, masked slots; the
    reference's bars against the restated oracle, the generic engine, run-to-run bit identity."""
    M, N, K, E, topk = shape
    inp = recipes.moe_int8_inputs(M, N, K, E, topk, 4100 + M)
    ids = inp["topk_ids"].clone()
    ids[::7, 0] = -1
    ref = moe.fused_experts_int8(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], inp["topk_weight"], ids).float()
    d = cuda(inp)
    idc = ids.cuda()
    w1p, w2p = ops.convert_weight_packed(d["w1"]), ops.convert_weight_packed(d["w2"])
    out = ops.fused_experts_cpu(d["a"].clone(), w1p, w2p, d["topk_weight"], idc, False, True, False, d["w1s"], d["w2s"],
                                None, None, None, True)
    mre = (out.float().cpu() - ref).abs().mean() / ref.abs().mean()
    assert mre < 0.01, f"mean relative error {mre:.4f}"
    assert ref_pred(ref, out)
    generic = ops.fused_experts_cpu(d["a"].clone(), d["w1"], d["w2"], d["topk_weight"], idc, False, True, False, d["w1s"],
                                    d["w2s"], None, None, None, False)
    assert rel_rms(out, generic) < 5e-3
    again = ops.fused_experts_cpu(d["a"].clone(), w1p, w2p, d["topk_weight"], idc, False, True, False, d["w1s"], d["w2s"],
                                  None, None, None, True)
    assert torch.equal(out, again), "run-to-run bit identity"


# ---- shared_expert ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", recipes.SHARED_CASES, ids=lambda c: c[0])
def test_shared_expert_bf16_and_int8(ops, case):
    name, m, n, k, rsf, seed = case
    g, _ = load_golden("shared_" + name)
    inp = cuda(recipes.shared_inputs(m, n, k, seed))
    hs = inp["hs"].clone()
    # 12-argument form of /root/reference/test_shared_experts.py:68
    res = ops.shared_expert_cpu(hs, inp["w1"], inp["w2"], inp["fused"], rsf, True, False, False, None, None, None, False)
    assert res.data_ptr() == hs.data_ptr() and ref_pred(g["ref_bf16"], res), name
    hs2 = inp["hs"].clone()
    res8 = ops.shared_expert_cpu(hs2, g["w1q"].cuda(), g["w2q"].cuda(), inp["fused"], rsf, True, True, False,
                                 g["w1s"].cuda(), g["w2s"].cuda(), None, False)
    assert ref_pred(g["ref_int8"], res8), name


@pytest.mark.parametrize("M", [129, 200, 500, 700, 1100, 1300])
def test_shared_expert_bf16_and_int8_packed_at_prefill_sizes(ops, knob, M):
    """Packed bf16 / int8 weights from 192 rows on (bf16: from SGLK_SHARED_MID_MAX, below it the split-K passes) run on the tuned
    256-row kernels of fused_experts -- grouped form with one expert for gate_up + SiLU*mul, dense form with the fused_out addend for
    down (the generic engine took 0.71 / 0.66 ms at 2048 x 2048 x 7168).  Same contract (/root/reference/test_shared_experts.py:34-53)
    against the fp32 / int8 oracle compositions, and within rounding of the generic engine (SGLK_FORCE_GENERIC=1)."""
    N, K, rsf = 512, 1024, 2.5
    g = torch.Generator().manual_seed(6100 + M)
    hs = (torch.randn(M, K, generator=g) / K ** 0.5).bfloat16()
    fused = (torch.randn(M, K, generator=g) * 0.2).bfloat16()       # MLP and addend of comparable size, |out| ~ 1
    w1 = torch.randn(2 * N, K, generator=g).bfloat16()
    w2 = (torch.randn(K, N, generator=g) / N ** 0.5).bfloat16()
    w1q = torch.randint(-127, 128, (2 * N, K), generator=g, dtype=torch.int8)
    w2q = torch.randint(-127, 128, (K, N), generator=g, dtype=torch.int8)
    w1s, w2s = torch.rand(2 * N, generator=g) * 2.7e-2, torch.rand(K, generator=g) * (2.7e-2 / N ** 0.5)
    d = [t.cuda() for t in (hs, fused, w1, w2, w1q, w2q, w1s, w2s)]
    p1, p2, q1, q2 = (ops.convert_weight_packed(t) for t in d[2:6])
    bf = lambda: ops.shared_expert_cpu(d[0], p1, p2, d[1], rsf, False, False, False, None, None, None, None, None, True)
    i8 = lambda: ops.shared_expert_cpu(d[0], q1, q2, d[1], rsf, False, True, False, d[6], d[7], None, None, None, True)
    out_b, out_i = bf(), i8()
    ref_b = moe.shared_expert_f32(hs, w1, w2, fused.float(), rsf)
    ref_i = moe.shared_expert_int8(hs, w1q, w2q, w1s, w2s, fused, rsf)
    assert ref_pred(ref_b, out_b) and rel_rms(out_b, ref_b) < 6e-3
    assert ref_pred(ref_i, out_i) and rel_rms(out_i, ref_i) < 6e-3
    # row-major weights (the reference's own 12-argument call passes is_vnni=False, test_shared_experts.py:68): re-tiled into the
    # workspace at these sizes, then the very same kernels -> the same bits
    rm_b = ops.shared_expert_cpu(d[0], d[2], d[3], d[1], rsf, False, False, False, None, None, None, False)
    rm_i = ops.shared_expert_cpu(d[0], d[4], d[5], d[1], rsf, False, True, False, d[6], d[7], None, False)
    assert torch.equal(rm_b, out_b) and torch.equal(rm_i, out_i)
    knob(SGLK_FORCE_GENERIC=1)
    assert rel_rms(out_b, bf().float().cpu()) < 4e-3
    assert rel_rms(out_i, i8().float().cpu()) < 4e-3


@pytest.mark.parametrize("prepack", [False, True])
@pytest.mark.parametrize("case", recipes.SHARED_FP8_CASES, ids=lambda c: c[0])
def test_shared_expert_fp8(ops, case, prepack):
    name, M, N, K, rsf, seed = case
    g, _ = load_golden("shared_fp8_" + name)
    inp = cuda(recipes.shared_fp8_inputs(M, N, K, seed))
    w1 = ops.convert_weight_packed(inp["w1"]) if prepack else inp["w1"]
    w2 = ops.convert_weight_packed(inp["w2"]) if prepack else inp["w2"]
    a2 = inp["a"].clone()
    # 14-argument form of /root/reference/test_moe_fp8_ext.py:60-61
    out = ops.shared_expert_cpu(a2, w1, w2, inp["fused"], rsf, True, False, True, inp["w1s"], inp["w2s"], [64, 128],
                                None, None, prepack)
    assert ref_pred(g["ref_out_f32"], out), name
    assert rel_rms(out, g["ref_out_f32"]) < 6e-3


@pytest.mark.parametrize("M", [192, 300, 700, 1000])
def test_shared_expert_fp8_between_decode_and_prefill_sizes(ops, knob, M):
    """192 <= M < 1024 runs as split-K passes of the weight-streaming kernel (the 256-row tile kernel has a handful of
    workgroups there: 0.19 -> 0.06 ms at 192 x 2048 x 7168): same contract (/root/reference/test_moe_fp8_ext.py:52-63) against
    the fp32 oracle composition, and within rounding of the tile-kernel path (SGLK_SHARED_MID_MAX=192)."""
    N, K, rsf, bn, bk = 256, 1024, 2.5, 128, 128
    inp = cuda(recipes.shared_fp8_inputs(M, N, K, 5300 + M, bn, bk))
    w1, w2 = ops.convert_weight_packed(inp["w1"]), ops.convert_weight_packed(inp["w2"])
    call = lambda: ops.shared_expert_cpu(inp["a"], w1, w2, inp["fused"], rsf, False, False, True, inp["w1s"], inp["w2s"], [bn, bk],
                                         None, None, True)
    out = call()
    W1 = moe.dequant_block_fp8(inp["w1"].cpu(), inp["w1s"].cpu(), bn, bk)
    W2 = moe.dequant_block_fp8(inp["w2"].cpu(), inp["w2s"].cpu(), bn, bk)
    ref = moe.shared_expert_f32(inp["a"].cpu(), W1, W2, inp["fused"].cpu().float(), rsf)
    assert ref_pred(ref, out)
    assert rel_rms(out, ref) < 6e-3
    rowmajor = ops.shared_expert_cpu(inp["a"], inp["w1"], inp["w2"], inp["fused"], rsf, False, False, True, inp["w1s"], inp["w2s"],
                                     [bn, bk], None, None, False)
    assert torch.equal(rowmajor, out), "row-major weights are re-tiled into the workspace and take the packed path"
    knob(SGLK_SHARED_MID_MAX=192)
    tile = call()
    assert rel_rms(out, tile.float().cpu()) < 4e-3


# ---- dense GEMMs -----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prepack", [False, True])
@pytest.mark.parametrize("case", recipes.GEMM_BF16_CASES, ids=lambda c: c[0])
def test_weight_packed_linear(ops, case, prepack):
    name, M, N, K, has_bias, seed = case
    g, _ = load_golden("gemm_bf16_" + name)
    inp = cuda(recipes.gemm_bf16_inputs(M, N, K, has_bias, seed))
    w = ops.convert_weight_packed(inp["mat2"]) if prepack else inp["mat2"]
    out = ops.weight_packed_linear(inp["mat1"], w, inp.get("bias"), prepack)
    assert out.dtype == torch.bfloat16 and ref_pred(g["ref_out"], out), name


def test_bf16_prepack_matches_reference_layout(ops):
    """The one layout the reference pins with a live known-answer test (/root/reference/test_gemm.py:36-46)."""
    oc, ic = 16 * 8, 32 * 24
    w1 = torch.randn(oc, ic, device="cuda").bfloat16()
    packed = ops.convert_weight_packed(w1)
    ref = w1.view(oc // 32, 32, ic // 2, 2).permute(0, 2, 1, 3).contiguous().view(oc, ic)
    assert torch.equal(ref, packed)


@pytest.mark.parametrize("prepack", [False, True])
@pytest.mark.parametrize("case", recipes.GEMM_FP8_CASES, ids=lambda c: c[0])
def test_fp8_scaled_mm(ops, case, prepack):
    name, M, N, K, has_bias, chunk, seed = case
    g, _ = load_golden("gemm_fp8_" + name)
    inp = recipes.gemm_fp8_inputs(M, N, K, has_bias, chunk, seed)
    data = inp["data"].cuda() if not chunk else torch.empty(M, K + 6, dtype=torch.bfloat16, device="cuda").narrow(1, 0, K).copy_(inp["data"])
    assert (data.stride(0) != K) == chunk
    w = inp["w"].cuda()
    w = ops.convert_weight_packed(w) if prepack else w
    bias = inp["bias"].cuda() if has_bias else None
    out = ops.fp8_scaled_mm_cpu(data, w, inp["scales"].cuda(), [64, 128], bias, data.dtype, prepack)
    assert ref_pred(g["ref_out_bf16"], out), name          # the reference's own (bf16) oracle
    assert rel_rms(out, g["ref_out_f32"]) < 4e-3
    out32 = ops.fp8_scaled_mm_cpu(data, w, inp["scales"].cuda(), [64, 128], bias, torch.float32, prepack)
    torch.testing.assert_close(out32.cpu(), g["ref_out_f32"], rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("name", ["m500_n768_k1024_bias", "m333_n512_k2048_chunk"])
def test_fp8_scaled_mm_goldens_on_the_128_token_kernel(ops, knob, name):
    """Dense fp8 W8A16 on moe_gemm_fp8w_s128.hip (MODE_PLAIN: the bf16 activations as two exact e4m3 terms on the block-scaled fp8
    matrix cores, 128-row tiles, two workgroups per CU; the default for launches of >= 1024 such tiles, here forced from 128 rows): the reference's
    golden cases incl. the bias and the row-strided `narrow` view (/root/reference/test_gemm_fp8.py:32-58), and within rounding of
    the 256-row bf16-MFMA kernel."""
    case = next(c for c in recipes.GEMM_FP8_CASES if c[0] == name)
    _, M, N, K, has_bias, chunk, seed = case
    g, _ = load_golden("gemm_fp8_" + name)
    inp = recipes.gemm_fp8_inputs(M, N, K, has_bias, chunk, seed)
    data = inp["data"].cuda() if not chunk else torch.empty(M, K + 10, dtype=torch.bfloat16, device="cuda").narrow(1, 0, K).copy_(inp["data"])
    w = ops.convert_weight_packed(inp["w"].cuda())
    bias = inp["bias"].cuda() if has_bias else None
    knob(SGLK_DENSE_S128=1)
    out = ops.fp8_scaled_mm_cpu(data, w, inp["scales"].cuda(), [64, 128], bias, data.dtype, True)
    assert ref_pred(g["ref_out_bf16"], out), name
    assert rel_rms(out, g["ref_out_f32"]) < 4e-3
    assert torch.equal(out, ops.fp8_scaled_mm_cpu(data, w, inp["scales"].cuda(), [64, 128], bias, data.dtype, True))
    knob(SGLK_DENSE_S128=0)
    out256 = ops.fp8_scaled_mm_cpu(data, w, inp["scales"].cuda(), [64, 128], bias, data.dtype, True)
    assert rel_rms(out, out256) < 3e-3


@pytest.mark.parametrize("shape", [(1000, 5120, 2048, True), (4096, 1536, 2048, False), (777, 2048, 6144, True), (2049, 512, 256, False)],
                         ids=lambda s: "x".join(map(str, s[:3])))
def test_fp8_scaled_mm_prefill_sizes_against_the_fp32_oracle(ops, knob, shape):
    """Qwen3 projection shapes at prefill sizes on the default path (the 128-token two-term kernel) and on the 256-row kernel
    (SGLK_DENSE_S128=0), against the fp32 oracle (oracle/gemm.py: /root/reference/test_gemm_fp8.py:32-45) with the reference
    predicate; a ragged last row tile, the shortest reduction (two K blocks) and a bias included."""
    M, N, K, has_bias = shape
    inp = recipes.gemm_fp8_inputs(M, N, K, has_bias, False, 4300 + M)
    bias = inp["bias"] if has_bias else None
    ref = ogemm.fp8_scaled_mm(inp["data"], inp["w"], inp["scales"], (64, 128), bias)
    w = ops.convert_weight_packed(inp["w"].cuda())
    b = bias.cuda() if has_bias else None
    for kn in (1, 0):     # 1: the 128-token kernel from 128 rows on (by default it takes launches of >= 1024 of its tiles)
        knob(SGLK_DENSE_S128=kn)
        out = ops.fp8_scaled_mm_cpu(inp["data"].cuda(), w, inp["scales"].cuda(), [64, 128], b, torch.bfloat16, True)
        assert ref_pred(ref, out), (shape, kn)
        assert rel_rms(out, ref) < 4e-3, (shape, kn)


@pytest.mark.parametrize("case", recipes.GEMM_INT8_CASES, ids=lambda c: c[0])
def test_int8_gemm_ops(ops, case):
    name, M, N, K, has_bias, seed = case
    g, _ = load_golden("gemm_int8_" + name)
    inp = cuda(recipes.gemm_int8_inputs(M, N, K, has_bias, seed))
    Aq, As = ops.per_token_quant_int8_cpu(inp["A"])
    assert torch.equal(Aq.cpu(), g["ref_Aq"]), "per-token int8 quantisation must be bit-exact"
    assert torch.equal(As.cpu(), g["ref_As"])
    bias = inp.get("bias")
    out = ops.int8_scaled_mm_cpu(Aq, inp["Bq"], As, inp["Bs"], bias, torch.bfloat16, False)
    assert ref_pred(g["ref_out"], out), name
    fused = ops.int8_scaled_mm_with_quant(inp["A"], inp["Bq"], inp["Bs"], bias, torch.bfloat16, False)
    assert torch.equal(fused, out), "fused quant+mm must equal the two-step path bit for bit"
    if N % 16 == 0 and K % 64 == 0:
        outp = ops.int8_scaled_mm_cpu(Aq, ops.convert_weight_packed(inp["Bq"]), As, inp["Bs"], bias, torch.bfloat16, True)
        if M >= 192 and N % 256 == 0 and K >= 256:
            # packed + large M runs on the int8 matrix cores with exact int32 sums (test_int8_mfma_gemm_is_exact); the
            # row-major path accumulates the same products in fp32, which rounds once sums pass 2^24
            assert ref_pred(g["ref_out"], outp), name
            assert rel_rms(outp, out) < 2e-3
        else:
            assert torch.equal(outp, out)


@pytest.mark.parametrize("shape", [(192, 256, 256, False), (1000, 512, 1024, True), (300, 768, 4160, True), (2049, 1536, 2048, False),
                                   (1, 256, 512, True), (64, 512, 1024, True), (128, 256, 4096, False), (37, 1536, 7168, True),
                                   # Qwen3-30B-A3B projection shapes of BASELINE config 1 (/root/reference/test_gemm_int8.py:66-73 on
                                   # "Qwen3 FFN shapes"): qkv [5120, 2048], o [2048, 4096], dense gate_up [12288, 2048], down [2048, 6144]
                                   (1000, 5120, 2048, True), (1000, 2048, 4096, False), (1000, 12288, 2048, False), (1000, 2048, 6144, True),
                                   (4096, 12288, 2048, True), (16, 5120, 2048, False)],
                         ids=lambda s: "x".join(map(str, s[:3])))
def test_int8_mfma_gemm_is_exact(ops, shape):
    """Packed int8 GEMMs run on the int8 matrix cores -- mfma_i32_32x32x32_i8 (csrc/gemm_i8_256.hip) for M >= 192, the
    weight-streaming mfma_i32_16x16x64_i8 kernel with exact int32 split-K partials (csrc/gemm_i8_mid.hip) for M <= 128 where the
    shape allows: the integer dot products are exact,
    so the result must equal an exact-integer evaluation of the oracle's expression (As * C * Bs + bias, fp32, one bf16
    rounding; /root/reference/test_gemm_int8.py:41-47) BIT FOR BIT, and the reference predicate against the float oracle."""
    M, N, K, has_bias = shape
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) / 10).bfloat16()
    Bq = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8)
    Bs = torch.rand(N, generator=g) * 1e-2 + 1e-4
    bias = torch.randn(N, generator=g) if has_bias else None
    Aq, As = ogemm.per_token_quant_int8(A)
    acc = (Aq.double() @ Bq.double().t()).long()                        # exact: |sum| <= 127 * 128 * K < 2^53
    exact = As.float().view(-1, 1) * acc.to(torch.float32) * Bs.view(1, -1)
    if bias is not None:
        exact = exact + bias.view(1, -1)
    exact = exact.bfloat16()
    wp = ops.convert_weight_packed(Bq.cuda())
    b = bias.cuda() if bias is not None else None
    out = ops.int8_scaled_mm_cpu(Aq.cuda(), wp, As.cuda(), Bs.cuda(), b, torch.bfloat16, True)
    assert torch.equal(out.cpu(), exact)
    fused = ops.int8_scaled_mm_with_quant(A.cuda(), wp, Bs.cuda(), b, torch.bfloat16, True)
    assert torch.equal(fused, out)
    assert ref_pred(ogemm.int8_scaled_mm(Aq, As, Bq, Bs, bias).bfloat16(), out)


@pytest.mark.parametrize("shape", [(129, 4096, 4096, True), (160, 2048, 6144, False), (256, 4096, 4096, False), (300, 384, 768, True),
                                   (512, 4096, 4096, True), (1000, 640, 1024, False)],
                         ids=lambda s: "x".join(map(str, s[:3])))
def test_int8_dense_between_129_and_1023_rows_is_exact(ops, knob, shape):
    """int8_scaled_mm_cpu / _with_quant above 128 rows while the 256-row kernel would have only a handful of workgroups: several
    128-row tiles of csrc/gemm_i8_mid.hip with exact int32 split-K partials.  BIT FOR BIT the exact-integer evaluation of the
    oracle's expression (/root/reference/test_gemm_int8.py:41-47), and bit-identical to the round-2 policy
    (SGLK_I8_DENSE_MID_WGS=0) where that ran the 256-row int8 kernel."""
    M, N, K, has_bias = shape
    g = torch.Generator().manual_seed(M + 2 * N + K)
    A = (torch.randn(M, K, generator=g) / 10).bfloat16()
    Bq = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8)
    Bs = torch.rand(N, generator=g) * 1e-2 + 1e-4
    bias = torch.randn(N, generator=g) if has_bias else None
    Aq, As = ogemm.per_token_quant_int8(A)
    acc = (Aq.double() @ Bq.double().t()).long()
    exact = As.float().view(-1, 1) * acc.to(torch.float32) * Bs.view(1, -1)
    if bias is not None:
        exact = exact + bias.view(1, -1)
    exact = exact.bfloat16()
    wp = ops.convert_weight_packed(Bq.cuda())
    b = bias.cuda() if bias is not None else None
    knob(SGLK_I8_DENSE_MID_WGS=100000)      # every shape of this test on the weight-streaming kernel
    out = ops.int8_scaled_mm_cpu(Aq.cuda(), wp, As.cuda(), Bs.cuda(), b, torch.bfloat16, True)
    assert torch.equal(out.cpu(), exact)
    fused = ops.int8_scaled_mm_with_quant(A.cuda(), wp, Bs.cuda(), b, torch.bfloat16, True)
    assert torch.equal(fused, out)
    knob(SGLK_I8_DENSE_MID_WGS=None)        # the shipped policy
    assert torch.equal(ops.int8_scaled_mm_cpu(Aq.cuda(), wp, As.cuda(), Bs.cuda(), b, torch.bfloat16, True), out)
    knob(SGLK_I8_DENSE_MID_WGS=0)           # round-2 policy: the generic engine up to 191 rows, the 256-row int8 kernel from 192 on
    old = ops.int8_scaled_mm_cpu(Aq.cuda(), wp, As.cuda(), Bs.cuda(), b, torch.bfloat16, True)
    if M >= 192 and N % 256 == 0:
        assert torch.equal(old, out)
    else:
        assert ref_pred(out.float().cpu(), old)


@pytest.mark.parametrize("shape", [(1024, 2048, 1024, True), (1500, 1024, 2048, False), (2047, 768, 512, True)],
                         ids=lambda s: "x".join(map(str, s[:3])))
def test_dense_gemms_between_1024_and_2047_rows_on_the_streaming_kernels(ops, knob, shape):
    """From 1024 to 2047 rows the dense GEMMs stay on the weight-streaming kernels while the 256-row kernels would have few
    workgroups (SGLK_DENSE_MID_MAX = 2048).  Every weight type against its oracle (bf16: fp32 matmul rounded once,
    /root/reference/test_gemm.py:15-21; fp8: oracle/gemm.py after /root/reference/test_gemm_fp8.py:32-45; int8: the exact-integer
    evaluation of /root/reference/test_gemm_int8.py:41-47, bit for bit) and against the tile kernels (SGLK_DENSE_MID_MAX=1024)."""
    M, N, K, has_bias = shape
    g = torch.Generator().manual_seed(M + 7 * N + K)
    x = (torch.randn(M, K, generator=g) / 8).bfloat16()
    wb = (torch.randn(N, K, generator=g) / 8).bfloat16()
    bias = torch.randn(N, generator=g) if has_bias else None
    b = bias.cuda() if has_bias else None
    ref_b = x.float() @ wb.float().t() + (bias if has_bias else 0)
    inp = recipes.gemm_fp8_inputs(M, N, K, has_bias, False, 5300 + M)
    ref_f = ogemm.fp8_scaled_mm(inp["data"], inp["w"], inp["scales"], (64, 128), inp["bias"] if has_bias else None)
    Bq = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8)
    Bs = torch.rand(N, generator=g) * 1e-2 + 1e-4
    Aq, As = ogemm.per_token_quant_int8(x)
    exact = As.float().view(-1, 1) * (Aq.double() @ Bq.double().t()).long().to(torch.float32) * Bs.view(1, -1)
    exact = (exact + bias.view(1, -1) if has_bias else exact).bfloat16()
    wbp, wfp, wip = (ops.convert_weight_packed(t.cuda()) for t in (wb, inp["w"], Bq))
    fb = inp["bias"].cuda() if has_bias else None
    outs = {}
    for mx in (None, 1024):
        knob(SGLK_DENSE_MID_MAX=mx, SGLK_DENSE_MID_WGS_BF16=100000 if mx is None else None,
             SGLK_DENSE_MID_WGS_FP8=100000 if mx is None else None, SGLK_I8_DENSE_MID_WGS=100000 if mx is None else None)
        ob = ops.weight_packed_linear(x.cuda(), wbp, b, True)
        of = ops.fp8_scaled_mm_cpu(inp["data"].cuda(), wfp, inp["scales"].cuda(), [64, 128], fb, torch.bfloat16, True)
        oi = ops.int8_scaled_mm_cpu(Aq.cuda(), wip, As.cuda(), Bs.cuda(), b, torch.bfloat16, True)
        assert ref_pred(ref_b, ob) and rel_rms(ob, ref_b) < 3e-3, (shape, mx)
        assert ref_pred(ref_f, of) and rel_rms(of, ref_f) < 4e-3, (shape, mx)
        assert torch.equal(oi.cpu(), exact), (shape, mx)
        outs[mx] = (ob, of)
    assert rel_rms(outs[None][0], outs[1024][0].float().cpu()) < 3e-3
    assert rel_rms(outs[None][1], outs[1024][1].float().cpu()) < 4e-3


@pytest.mark.parametrize("shape", [(192, 256, 128, False), (1000, 512, 1024, True), (300, 768, 2080, True), (2049, 1536, 2048, False)],
                         ids=lambda s: "x".join(map(str, s[:3])))
def test_bf16_packed_linear_on_tuned_kernel(ops, shape):
    """Large-M weight_packed_linear with VNNI-2 packed weights runs on csrc/gemm_bf16_256.hip; oracle: fp32 matmul of the
    bf16 operands (+ bias) rounded once (/root/reference/test_gemm.py:15-21), reference predicate; also equal within
    rounding noise to the row-major (generic engine) path."""
    M, N, K, has_bias = shape
    g = torch.Generator().manual_seed(M * 3 + N + K)
    x = (torch.randn(M, K, generator=g) / 8).bfloat16()
    w = (torch.randn(N, K, generator=g) / 8).bfloat16()
    bias = torch.randn(N, generator=g) if has_bias else None
    ref = x.float() @ w.float().t()
    if bias is not None:
        ref = ref + bias
    wp = ops.convert_weight_packed(w.cuda())
    b = bias.cuda() if bias is not None else None
    out = ops.weight_packed_linear(x.cuda(), wp, b, True)
    assert ref_pred(ref, out)
    assert rel_rms(out, ref) < 3e-3
    plain = ops.weight_packed_linear(x.cuda(), w.cuda(), b, False)
    assert rel_rms(out, plain) < 3e-3


@pytest.mark.parametrize("shape", [(128, 4096, 4096, False), (65, 512, 1024, True), (100, 384, 768, True), (128, 256, 384, False),
                                   (160, 2048, 1024, True), (97, 12288, 2048, False)],
                         ids=lambda s: "x".join(map(str, s[:3])))
def test_bf16_packed_linear_between_65_and_191_rows(ops, shape):
    """weight_packed_linear with packed weights at 65 ... 191 rows (BASELINE config 0 is (128, 4096, 4096),
    /root/reference/test_gemm.py:51-72) runs csrc/gemm_bf16_mid.hip with 128-row tiles and split-K; oracle: fp32 matmul of
    the bf16 operands (+ bias) rounded once (/root/reference/test_gemm.py:15-21), reference predicate; and within rounding
    noise of the row-major (generic engine) path."""
    M, N, K, has_bias = shape
    g = torch.Generator().manual_seed(M * 5 + N + K)
    x = (torch.randn(M, K, generator=g) / 8).bfloat16()
    w = (torch.randn(N, K, generator=g) / 8).bfloat16()
    bias = torch.randn(N, generator=g) if has_bias else None
    ref = x.float() @ w.float().t()
    if bias is not None:
        ref = ref + bias
    wp = ops.convert_weight_packed(w.cuda())
    b = bias.cuda() if bias is not None else None
    out = ops.weight_packed_linear(x.cuda(), wp, b, True)
    assert ref_pred(ref, out)
    assert rel_rms(out, ref) < 3e-3
    plain = ops.weight_packed_linear(x.cuda(), w.cuda(), b, False)
    assert rel_rms(out, plain) < 3e-3


@pytest.mark.parametrize("shape", [(1024, 768, 2048, 16, 4), (1000, 128, 256, 4, 2), (3000, 384, 1024, 32, 8)],
                         ids=lambda s: "x".join(map(str, s)))
def test_fused_experts_bf16_on_tuned_kernel(ops, shape):
    """Large-M packed bf16 fused_experts runs both grouped GEMMs on csrc/gemm_bf16_256.hip; oracle: the fp32 restatement
    of /root/reference/test_moe.py:22-54 (oracle/moe.py), reference predicate; masked slots; run-to-run bit identity."""
    M, N, K, E, topk = shape
    g = torch.Generator().manual_seed(5000 + M)
    a = (torch.randn(M, K, generator=g) / 10).bfloat16()
    w1 = (torch.randn(E, 2 * N, K, generator=g) / 10).bfloat16()
    w2 = (torch.randn(E, K, N, generator=g) / 10).bfloat16()
    tw, ids = moe.softmax_topk(torch.randn(M, E, generator=g).bfloat16(), topk, True)
    ids = ids.clone()
    ids[::5, 0] = -1
    ref = moe.fused_experts_f32(a, w1, w2, tw, ids)
    w1p, w2p = ops.convert_weight_packed(w1.cuda()), ops.convert_weight_packed(w2.cuda())
    args = (tw.cuda(), ids.cuda(), False, False, False, None, None, None, None, None)
    out = ops.fused_experts_cpu(a.cuda(), w1p, w2p, *args, True)
    assert rel_rms(out, ref) < 6e-3
    assert ref_pred(ref, out) or rel_rms(out, ref) < 4e-3   # |out| reaches O(10) here: the absolute part of the predicate is tight
    generic = ops.fused_experts_cpu(a.cuda(), w1.cuda(), w2.cuda(), *args, False)
    assert rel_rms(out, generic) < 4e-3
    again = ops.fused_experts_cpu(a.cuda(), w1p, w2p, *args, True)
    assert torch.equal(out, again)


@pytest.mark.parametrize("shape", [(200, 768, 2048, 16, 4), (4, 384, 1024, 32, 8), (150, 384, 640, 8, 2), (61, 256, 4352, 4, 2)],
                         ids=lambda s: "x".join(map(str, s)))
def test_fused_experts_bf16_on_mid_kernel(ops, shape):
    """Small / mid-size packed bf16 fused_experts (below ~72 rows per expert) runs on the weight-streaming kernel
    csrc/gemm_bf16_mid.hip: even, odd (N = 384 -> 3, K = 640 -> 5) and long (K = 4352 -> 34) K-block counts, masked slots,
    against the fp32 restatement of /root/reference/test_moe.py:22-54 and the generic engine; run-to-run bit identity."""
    M, N, K, E, topk = shape
    g = torch.Generator().manual_seed(6000 + M)
    a = (torch.randn(M, K, generator=g) / 10).bfloat16()
    w1 = (torch.randn(E, 2 * N, K, generator=g) / 10).bfloat16()
    w2 = (torch.randn(E, K, N, generator=g) / 10).bfloat16()
    tw, ids = moe.softmax_topk(torch.randn(M, E, generator=g).bfloat16(), topk, True)
    ids = ids.clone()
    ids[::5, 0] = -1
    ref = moe.fused_experts_f32(a, w1, w2, tw, ids)
    w1p, w2p = ops.convert_weight_packed(w1.cuda()), ops.convert_weight_packed(w2.cuda())
    args = (tw.cuda(), ids.cuda(), False, False, False, None, None, None, None, None)
    out = ops.fused_experts_cpu(a.cuda(), w1p, w2p, *args, True)
    assert rel_rms(out, ref) < 6e-3
    assert ref_pred(ref, out) or rel_rms(out, ref) < 4e-3
    generic = ops.fused_experts_cpu(a.cuda(), w1.cuda(), w2.cuda(), *args, False)
    assert rel_rms(out, generic) < 4e-3
    again = ops.fused_experts_cpu(a.cuda(), w1p, w2p, *args, True)
    assert torch.equal(out, again)


def test_fp8_generic_engine_matches_tuned_kernels(ops, knob):
    """Same fp8 fused_experts inputs through the tuned path and (forced) through the generic engine."""
    name, M, N, K, E, topk, bn, bk, masked, seed, full = recipes.MOE_FP8_CASES[1]
    g, _ = load_golden("moe_fp8_" + name)
    inp = cuda(recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, masked, seed))
    w1p, w2p = ops.convert_weight_packed(inp["w1"]), ops.convert_weight_packed(inp["w2"])
    args = (inp["topk_weight"], inp["topk_ids"], False, False, True, inp["w1s"], inp["w2s"], [bn, bk], None, None)
    tuned = ops.fused_experts_cpu(inp["a"], w1p, w2p, *args, True)
    knob(SGLK_FORCE_GENERIC=1)
    generic_packed = ops.fused_experts_cpu(inp["a"], w1p, w2p, *args, True)
    generic_plain = ops.fused_experts_cpu(inp["a"], inp["w1"], inp["w2"], *args, False)     # is_vnni=False
    assert torch.equal(generic_packed, generic_plain)
    for o in (tuned, generic_packed):
        assert ref_pred(g["ref_out_f32"], o)
    assert rel_rms(tuned, generic_packed) < 5e-3
