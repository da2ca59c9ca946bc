"""The opt-in a8 mode of fp8 fused_experts (fp8 activations on the block-scaled fp8 matrix cores; sglk.h SGLK_MOE_FP8_ACT).

NOT the reference's numerics: /root/reference/bench_moe.py:113-130 is W8A16.  So these tests hold the kernels to an oracle
of THEIR arithmetic (oracle/moe_a8.py: quantise exactly as the kernels do, then exact sums) at a stated tolerance --
relative RMS < 5e-3 (measured ~1e-3: the bf16 roundings of ic2 / out plus rare rounding-boundary flips of the fp8
quantisation) -- and only REPORT how far the mode is from the reference's W8A16 oracle (about 4-5 % relative RMS; it does
not meet the reference predicate and is therefore never a default).  The quantisation pass itself is bit-exact.
"""
import ctypes

import pytest
import torch

import recipes
from oracle import c_oracle, moe_a8

pytestmark = pytest.mark.gpu

A8_TOL = 5e-3


@pytest.fixture(scope="module")
def ops():
    import sgl_kernel  # noqa: F401
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.ops.sgl_kernel


@pytest.fixture
def a8():
    from sgl_kernel import _ops
    _ops.set_fp8_activations(True)
    yield _ops
    _ops.set_fp8_activations(False)


def gpu_quant(x):
    from sgl_kernel import _lib
    rows, cols = x.shape
    q = torch.empty(rows, cols, dtype=torch.uint8, device="cuda")
    ss = (cols // 128 + 3) // 4 * 4
    s = torch.zeros(rows, ss, dtype=torch.uint8, device="cuda")
    rc = _lib.lib().sglk_quant_fp8_block128(ctypes.c_void_p(x.data_ptr()), x.stride(0), ctypes.c_void_p(q.data_ptr()), cols,
                                            ctypes.c_void_p(s.data_ptr()), ss, rows, cols,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "quant_fp8_block128")
    torch.cuda.synchronize()
    return q.cpu(), s.cpu()[:, :cols // 128]


@pytest.mark.parametrize("rows,cols", [(1, 128), (37, 2048), (5, 768), (3, 4096 + 128)])
def test_quant_fp8_block128_is_bit_exact(ops, rows, cols):
    """e4m3 bytes and E8M0 scale bytes equal the oracle's, position by position (after undoing the packed-tile k order);
    rows with zeros, denormals, huge values, and block maxima exactly on / next to the 1.75 * 2^k scale boundary."""
    g = torch.Generator().manual_seed(rows * 1000 + cols)
    x = torch.randn(rows, cols, generator=g) * torch.exp2(torch.randint(-20, 12, (rows, 1), generator=g).float())
    x[0, :128] = 0.0
    if rows > 2:
        x[1, 0], x[1, 1], x[1, 2] = 1.75, -1.7578125, 3.5          # bf16-exact values around the boundary mantissa
        x[2, :128] = 1e-39                                        # bf16 denormals
        x[2, 128 % cols:(128 % cols) + 4] = torch.tensor([3e38, -3e38, 448.0, -449.0])[: min(4, cols - 128 % cols)]
    xb = x.bfloat16()
    q, s = gpu_quant(xb.cuda())
    _, q_ref, sb_ref = moe_a8.quant_block128(xb.float())
    order = torch.from_numpy(moe_a8.packed_k_order(cols))
    assert torch.equal(s.int(), sb_ref), "E8M0 scale bytes differ"
    assert torch.equal(q, q_ref[:, order]), "e4m3 bytes differ"


def run(ops, inp, block):
    d = {k: v.cuda() for k, v in inp.items()}
    w1p, w2p = ops.convert_weight_packed(d["w1"]), ops.convert_weight_packed(d["w2"])
    out = ops.fused_experts_cpu(d["a"], w1p, w2p, d["topk_weight"], d["topk_ids"], False, False, True,
                                d["w1s"], d["w2s"], list(block), None, None, True)
    torch.cuda.synchronize()
    return out


def check_a8(out, ref_q, ref_w8a16, what):
    got = out.float().cpu()
    rel_q = ((got - ref_q).norm() / ref_q.norm().clamp_min(1e-12)).item()
    rel = ((got - ref_w8a16).norm() / ref_w8a16.norm().clamp_min(1e-12)).item()
    pred = torch.allclose(ref_w8a16.bfloat16(), out.cpu(), rtol=1e-2, atol=1e-2)
    print(f"[a8] {what}: rel RMS vs quantised oracle {rel_q:.2e} (bound {A8_TOL}); vs the reference's W8A16 oracle {rel:.2e}, "
          f"reference predicate {'met' if pred else 'NOT met'} (reported, not asserted)")
    assert torch.isfinite(got).all()
    assert rel_q < A8_TOL, f"{what}: a8 kernels vs their own oracle: relative RMS {rel_q:.2e}"
    return rel_q, rel


@pytest.mark.parametrize("name", ["m1212_n512_k1024_e8_t2", "masked_m300_n256_k512_e16_t8", "qwen3dims_m96_e8_t8", "m2_n128_k128_e8_t4"])
def test_fused_experts_a8_small_shapes(ops, a8, name):
    from sgl_kernel import _lib
    case = next(c for c in recipes.MOE_FP8_CASES if c[0] == name)
    _, M, N, K, E, topk, bn, bk, masked, seed, _full = case
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, masked, seed)
    if K % 256 != 0:
        with pytest.raises(RuntimeError, match="SGLK_MOE_FP8_ACT"):
            run(ops, inp, (bn, bk))        # an explicit request the kernels cannot take is refused, not silently re-routed
        return
    out = run(ops, inp, (bn, bk))
    assert a8.last_path & _lib.PATH_FP8_ACT
    args = (inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], (bn, bk), inp["topk_weight"], inp["topk_ids"])
    check_a8(out, moe_a8.fused_experts_a8(*args), c_oracle.fused_experts_fp8(*args), name)
    again = run(ops, inp, (bn, bk))
    assert torch.equal(out, again), "a8 mode must be run-to-run bit identical (no float atomics)"


# deviation of the a8 mode from the REFERENCE's own fp32 oracle on the reference's golden cases (tests/golden/moe_fp8_*.safetensors:
# `ref_out_f32` = native_fused_moe of /root/reference/test_moe_fp8_ext.py:70-91 run in the build container), measured once on
# the MI355X and committed here: (relative RMS, fraction of elements outside allclose(rtol = atol = 1e-2)).  The mode does NOT
# meet the reference predicate -- these numbers are what a caller who opts in accepts.
A8_VS_REFERENCE = {
    "m1212_n512_k1024_e8_t2": (0.0463, 0.0449),
    "m121_n512_k1024_e8_t2": (0.0465, 0.0459),
    "masked_m300_n256_k512_e16_t8": (0.0459, 0.2594),
    "qwen3dims_m96_e8_t8": (0.0458, 0.0891),
}


@pytest.mark.parametrize("name", sorted(A8_VS_REFERENCE))
def test_a8_deviation_from_the_reference_oracle_is_pinned(ops, a8, name):
    from conftest import load_golden
    case = next(c for c in recipes.MOE_FP8_CASES if c[0] == name)
    _, M, N, K, E, topk, bn, bk, masked, seed, _full = case
    g, _ = load_golden("moe_fp8_" + name)
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, masked, seed)
    out = run(ops, inp, (bn, bk)).float().cpu()
    ref = g["ref_out_f32"].float()
    rel = float((out - ref).norm() / ref.norm())
    outside = float((~torch.isclose(out.bfloat16().float(), ref.bfloat16().float(), rtol=1e-2, atol=1e-2)).float().mean())
    print(f"[a8 vs reference oracle] {name}: relative RMS {rel:.4f}, fraction outside allclose(1e-2) {outside:.4f}")
    want_rel, want_out = A8_VS_REFERENCE[name]
    if want_rel is not None:
        assert abs(rel - want_rel) <= 0.1 * want_rel + 1e-4, f"{name}: relative RMS {rel:.4f}, committed {want_rel}"
        assert abs(outside - want_out) <= 0.1 * want_out + 2e-3, f"{name}: fraction outside {outside:.4f}, committed {want_out}"
    assert rel < 0.08 and torch.isfinite(out).all()


def test_fused_experts_a8_long_reduction(ops, a8):
    """56 K blocks in GEMM-1 (K = 7168, the reference bench's hidden size, bench_moe.py:144) and three in GEMM-2: the 64-block scale
    tables of the 128-token kernel and the shortest odd rotation of its fragment sets."""
    M, N, K, E, topk, bn, bk = 600, 384, 7168, 8, 2, 128, 128
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, False, 9107)
    args = (inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], (bn, bk), inp["topk_weight"], inp["topk_ids"])
    ref_q = moe_a8.fused_experts_a8(*args)
    k = float(2.0 / ref_q.abs().max())
    inp["topk_weight"] = inp["topk_weight"] * k
    args = (inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], (bn, bk), inp["topk_weight"], inp["topk_ids"])
    out = run(ops, inp, (bn, bk))
    check_a8(out, moe_a8.fused_experts_a8(*args), c_oracle.fused_experts_fp8(*args), "K = 7168")


def test_fused_experts_a8_scale_extremes(ops, a8):
    """Block scales over 2^-12 .. 2^4 with random sign, a zero block, ragged expert loads (rows per expert far from 256)."""
    M, N, K, E, topk, bn, bk = 1531, 256, 512, 8, 4, 128, 128
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, False, 9001)
    g = torch.Generator().manual_seed(5)
    inp["w1s"] = inp["w1s"].sign() * torch.exp2(torch.rand(inp["w1s"].shape, generator=g) * 16 - 12) * 1e-2
    inp["w1s"][0, 0, 0] = 0.0
    inp["w2s"][1, 0, 0] = 0.0
    inp["w2s"][2, 1, 1] = 2.0 ** -9
    args = (inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], (bn, bk), inp["topk_weight"], inp["topk_ids"])
    ref_q = moe_a8.fused_experts_a8(*args)
    k = float(2.0 / ref_q.abs().max())
    inp["topk_weight"] = inp["topk_weight"] * k
    args = (inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], (bn, bk), inp["topk_weight"], inp["topk_ids"])
    out = run(ops, inp, (bn, bk))
    check_a8(out, moe_a8.fused_experts_a8(*args), c_oracle.fused_experts_fp8(*args), "scale extremes")


@pytest.mark.parametrize("M", [1000, 4096, 16384])
def test_fused_experts_a8_qwen3_full_experts(ops, a8, M):
    """Qwen3-30B-A3B expert dims, all 128 experts; sampled tokens against the quantised-arithmetic oracle."""
    from sgl_kernel import _lib
    N, K, E, topk, bn, bk = 768, 2048, 128, 8, 128, 128
    g = torch.Generator(device="cuda").manual_seed(777)
    w1 = (torch.randn(E, 2 * N, K, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn)
    w2 = (torch.randn(E, K, N, device="cuda", generator=g) * 400).clamp(-400, 400).to(torch.float8_e4m3fn)
    w1s = torch.randn(E, 2 * N // bn, K // bk, device="cuda", generator=g) * 1e-3
    w2s = torch.randn(E, K // bn, N // bk, device="cuda", generator=g) * 1e-3
    a = (torch.randn(M, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    score = torch.softmax(torch.randn(M, E, device="cuda", generator=g).bfloat16(), dim=-1, dtype=torch.float32)
    tw, ids = torch.topk(score, topk)
    ids = ids.to(torch.int32)
    w1p, w2p = ops.convert_weight_packed(w1), ops.convert_weight_packed(w2)
    out = ops.fused_experts_cpu(a, w1p, w2p, tw, ids, False, False, True, w1s, w2s, [bn, bk], None, None, True)
    torch.cuda.synchronize()
    assert a8.last_path & _lib.PATH_FP8_ACT
    sample = torch.arange(0, M, max(1, M // 48))[:48]
    args = (a[sample].cpu(), w1.cpu(), w2.cpu(), w1s.cpu(), w2s.cpu(), (bn, bk), tw[sample].cpu(), ids[sample].cpu())
    check_a8(out[sample], moe_a8.fused_experts_a8(*args), c_oracle.fused_experts_fp8(*args), f"qwen3 M={M}")
