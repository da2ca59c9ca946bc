"""fused_moe_block: router -> routed experts (-> shared expert) in one call.

The reference harness makes the calls one after the other -- grouped_topk_cpu + fused_experts_cpu
(/root/reference/test_moe.py:57-92), shared_expert_cpu on the routed output (/root/reference/test_shared_experts.py:34-40,68;
/root/reference/test_moe_fp8_ext.py:52-61) -- so parity here is: routing ids / weights bit-identical to the stand-alone
operator, the output within the reference's predicate of the golden outputs / the oracle composition, and equal to the
separate calls except for the one bf16 rounding of the routed output that the folded form does not make.
"""
import pytest
import torch

import recipes
from conftest import load_golden
from oracle import moe

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import sgl_kernel  # noqa: F401
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.ops.sgl_kernel


def ref_pred(ref, out):
    return torch.allclose(ref.to(out.dtype).cpu(), out.cpu(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("case", [c for c in recipes.MOE_BF16_CASES], ids=lambda c: c[0])
def test_block_bf16_matches_reference_flow(ops, case):
    """/root/reference/test_moe.py:57-92 as one call: ids / weights == grouped_topk_cpu's, output within the reference's
    predicate of the golden (reference oracle) output."""
    from sgl_kernel import _lib, _ops
    name, M, N, K, E, topk, renorm, seed, _full = case
    g, _ = load_golden("moe_bf16_" + name)
    inp = {k: v.cuda() for k, v in recipes.moe_bf16_inputs(M, N, K, E, topk, seed).items()}
    w1p, w2p = ops.convert_weight_packed(inp["w1"]), ops.convert_weight_packed(inp["w2"])
    tw_ref, ids_ref = ops.grouped_topk_cpu(inp["a"], inp["score"], topk, renorm, 1, 1, 0, None, None)
    out, tw, ids = ops.fused_moe_block(inp["a"], inp["score"], w1p, w2p, topk, renorm, 1, 1, None, False, False, False,
                                       None, None, None, True, None, None, None, None, 1.0)
    assert torch.equal(ids, ids_ref) and torch.equal(tw, tw_ref), "router inside the block != grouped_topk_cpu"
    if M <= 16:
        assert _ops.last_path & _lib.PATH_ROUTE_ALIGN, "decode-size batch must take the one-launch router + align"
    assert ref_pred(g["ref_out"], out), name
    sep = ops.fused_experts_cpu(inp["a"], w1p, w2p, tw_ref, ids_ref, False, False, False, None, None, None, None, None, True)
    assert torch.equal(out, sep), "without a shared expert the block is the separate calls, bit for bit"


@pytest.mark.parametrize("M,E,G,tg,biased", [(1, 128, 1, 1, False), (7, 128, 1, 1, False), (16, 256, 8, 4, False),
                                             (13, 256, 8, 2, True), (64, 128, 1, 1, False), (300, 128, 1, 1, False)])
def test_block_routing_is_bit_identical_and_sorted(ops, knob, M, E, G, tg, biased):
    """Router + align in one launch (M <= 16) against the two stand-alone launches: same ids, same weights, same output;
    larger M takes the unfused route inside the same entry point."""
    from sgl_kernel import _lib, _ops
    N, K, topk, bn, bk = 256, 512, 8, 128, 128
    inp = {k: v.cuda() for k, v in recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, False, 4000 + M).items()}
    g = torch.Generator(device="cuda").manual_seed(M)
    logits = torch.randn(M, E, device="cuda", generator=g).bfloat16()
    bias = torch.randn(E, device="cuda", generator=g).bfloat16() if biased else None
    w1p, w2p = ops.convert_weight_packed(inp["w1"]), ops.convert_weight_packed(inp["w2"])
    if biased:
        tw_ref = torch.empty(M, topk, dtype=torch.float32, device="cuda")
        ids_ref = torch.empty(M, topk, dtype=torch.int32, device="cuda")
        ops.biased_grouped_topk_cpu(tw_ref, ids_ref, inp["a"], logits, bias, topk, True, G, tg)
    else:
        tw_ref, ids_ref = ops.grouped_topk_cpu(inp["a"], logits, topk, True, G, tg, 0, None, None)
    out, tw, ids = ops.fused_moe_block(inp["a"], logits, w1p, w2p, topk, True, G, tg, bias, False, False, True,
                                       inp["w1s"], inp["w2s"], [bn, bk], True, None, None, None, None, 1.0)
    assert bool(_ops.last_path & _lib.PATH_ROUTE_ALIGN) == (M <= 16)
    assert torch.equal(ids, ids_ref) and torch.equal(tw, tw_ref)
    sep = ops.fused_experts_cpu(inp["a"], w1p, w2p, tw_ref, ids_ref, False, False, True, inp["w1s"], inp["w2s"], [bn, bk],
                                None, None, True)
    assert torch.equal(out, sep)
    ref = moe.fused_experts_fp8(inp["a"].cpu(), inp["w1"].cpu(), inp["w2"].cpu(), inp["w1s"].cpu(), inp["w2s"].cpu(), (bn, bk),
                                tw_ref.cpu(), ids_ref.cpu())
    assert ref_pred(ref, out)


@pytest.mark.parametrize("M", [1, 16, 64, 121, 300])
def test_block_with_shared_expert_fp8(ops, knob, M):
    """Routed experts + shared expert (DeepSeek-style block; /root/reference/test_moe_fp8_ext.py:27-63 is the shared half):
    against the oracle composition with the reference's predicate, and against the three separate operator calls."""
    from sgl_kernel import _lib, _ops
    N, Ns, K, E, topk, bn, bk, rsf = 256, 512, 1024, 16, 4, 128, 128, 2.5
    inp = {k: v.cuda() for k, v in recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, False, 5000 + M).items()}
    sh = {k: v.cuda() for k, v in recipes.shared_fp8_inputs(M, Ns, K, 6000 + M, bn, bk).items()}
    g = torch.Generator(device="cuda").manual_seed(M + 9)
    logits = torch.randn(M, E, device="cuda", generator=g).float()
    w1p, w2p = ops.convert_weight_packed(inp["w1"]), ops.convert_weight_packed(inp["w2"])
    s1p, s2p = ops.convert_weight_packed(sh["w1"]), ops.convert_weight_packed(sh["w2"])
    a = inp["a"]
    out, tw, ids = ops.fused_moe_block(a, logits, w1p, w2p, topk, True, 1, 1, None, False, False, True, inp["w1s"], inp["w2s"],
                                       [bn, bk], True, s1p, s2p, sh["w1s"], sh["w2s"], rsf)
    folded = bool(_ops.last_path & _lib.PATH_SHARED_FOLDED)
    assert folded == (M <= 192), f"M={M}: shared-expert fold {folded}"
    # the three separate calls of the reference flow
    tw_ref, ids_ref = ops.grouped_topk_cpu(a, logits, topk, True, 1, 1, 0, None, None)
    assert torch.equal(ids, ids_ref) and torch.equal(tw, tw_ref)
    routed = ops.fused_experts_cpu(a, w1p, w2p, tw_ref, ids_ref, False, False, True, inp["w1s"], inp["w2s"], [bn, bk], None, None, True)
    sep = ops.shared_expert_cpu(a, s1p, s2p, routed, rsf, False, False, True, sh["w1s"], sh["w2s"], [bn, bk], None, None, True)
    # oracle composition in fp32
    r32 = moe.fused_experts_fp8(a.cpu(), inp["w1"].cpu(), inp["w2"].cpu(), inp["w1s"].cpu(), inp["w2s"].cpu(), (bn, bk),
                                tw_ref.cpu(), ids_ref.cpu())
    W1 = moe.dequant_block_fp8(sh["w1"].cpu(), sh["w1s"].cpu(), bn, bk)
    W2 = moe.dequant_block_fp8(sh["w2"].cpu(), sh["w2s"].cpu(), bn, bk)
    ref = moe.shared_expert_f32(a.cpu(), W1, W2, r32, rsf)
    assert ref_pred(ref, out), f"M={M}: block vs oracle composition"
    err = (out.float().cpu() - ref).norm() / ref.norm()
    err_sep = (sep.float().cpu() - ref).norm() / ref.norm()
    assert err < 6e-3 and err <= err_sep * 1.05 + 1e-4, f"M={M}: block {err:.2e} vs separate calls {err_sep:.2e}"
    if not folded:
        assert torch.equal(out, sep), "unfolded shapes run the separate kernels: identical bits"
    # the A/B knob runs the block unfused: must equal the separate calls bit for bit
    knob(SGLK_NO_BLOCK_FOLD=1)
    out_nf, _, _ = ops.fused_moe_block(a, logits, w1p, w2p, topk, True, 1, 1, None, False, False, True, inp["w1s"], inp["w2s"],
                                       [bn, bk], True, s1p, s2p, sh["w1s"], sh["w2s"], rsf)
    assert not (_ops.last_path & _lib.PATH_SHARED_FOLDED)
    assert torch.equal(out_nf, sep)


def test_block_inplace_and_cpu_staging(ops):
    """inplace=True overwrites hidden_states (the routed and the shared GEMM-1 both read it before anything is written); CPU
    tensors are staged through the GPU like every operator of the package."""
    M, N, Ns, K, E, topk, bn, bk = 9, 256, 256, 512, 8, 2, 128, 128
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, False, 77)
    sh = recipes.shared_fp8_inputs(M, Ns, K, 78, bn, bk)
    logits = torch.randn(M, E, generator=torch.Generator().manual_seed(3))
    w1p, w2p = ops.convert_weight_packed(inp["w1"]), ops.convert_weight_packed(inp["w2"])
    s1p, s2p = ops.convert_weight_packed(sh["w1"]), ops.convert_weight_packed(sh["w2"])
    a = inp["a"].clone()
    ref, tw0, ids0 = ops.fused_moe_block(inp["a"].cuda(), logits.cuda(), w1p.cuda(), w2p.cuda(), topk, False, 1, 1, None, False, False,
                                         True, inp["w1s"].cuda(), inp["w2s"].cuda(), [bn, bk], True, s1p.cuda(), s2p.cuda(),
                                         sh["w1s"].cuda(), sh["w2s"].cuda(), 1.0)
    out, tw, ids = ops.fused_moe_block(a, logits, w1p, w2p, topk, False, 1, 1, None, True, False, True, inp["w1s"], inp["w2s"],
                                       [bn, bk], True, s1p, s2p, sh["w1s"], sh["w2s"], 1.0)
    assert out.device.type == "cpu" and out.data_ptr() == a.data_ptr()
    assert torch.equal(out, ref.cpu()) and torch.equal(ids, ids0.cpu()) and torch.equal(tw, tw0.cpu())
