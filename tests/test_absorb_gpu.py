"""GPU parity of qkv_proj_with_rope (MLA absorbed projection + RMSNorm + RoPE), through torch.ops.sgl_kernel -> C-ABI -> HIP.
Expected values: tests/golden/absorb_* = outputs of the reference's own oracles native_torch / native_torch_int8
(/root/reference/test_absorb.py:65-109) on the seeded inputs of tests/recipes.py; comparisons are the reference's
(test_absorb.py:149-153,191-194: utils.compare on q_input, its rotary slice, k_input, v_input)."""
import pytest
import torch

import recipes
from conftest import load_golden
from oracle.gemm import quant_int8_rowwise

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import sgl_kernel  # noqa: F401
    assert torch.cuda.is_available()
    return torch.ops.sgl_kernel


def ref_pred(ref, out):
    """utils.compare's predicate (/root/reference/utils.py:9-13) on bf16."""
    return torch.allclose(ref.bfloat16().cpu(), out.bfloat16().cpu(), rtol=1e-2, atol=1e-2)


def close(ref, out):
    """Parity bar of this operator.  The reference's compare() only PRINTS allclose(1e-2) (utils.py:14-25), and for this
    op that predicate is not meaningful element by element: q after q_b_proj is O(10) in bf16 (ulp 0.0625-0.125), and the
    rotation and the w_kc product subtract such numbers, so ONE differently rounded bf16 intermediate -- any fp32
    summation order differs from torch's -- moves a small output by several 1e-2 (the reference's own bf16-vs-int8
    comparison, test_absorb.py:173-177, prints False for the same reason).  Stated tolerance instead: relative RMS error
    < 5e-3 and every element within 3 bf16 ulps of the tensor's largest magnitude."""
    return within(ref, out, 5e-3, 3)


def within(ref, out, rms_tol, ulps):
    r, o = ref.float().cpu(), out.float().cpu()
    rms = float((o - r).norm() / r.norm().clamp_min(1e-12))
    big = float(r.abs().max())
    ulp = 2.0 ** (torch.tensor(big).log2().floor().item() - 7)
    return rms < rms_tol and float((o - r).abs().max()) <= ulps * ulp


@pytest.mark.parametrize("prepack", [True, False])
@pytest.mark.parametrize("case", recipes.ABSORB_CASES, ids=lambda c: c[0])
def test_qkv_proj_with_rope_bf16(ops, case, prepack):
    name, B, hidden, seed = case
    g, _ = load_golden("absorb_" + name)
    d = {k: v.cuda() for k, v in recipes.absorb_inputs(B, hidden, seed).items()}
    pk = ops.convert_weight_packed if prepack else (lambda w: w)
    q, k, v = ops.qkv_proj_with_rope(d["hidden_states"], pk(d["q_a_proj_weight"]), pk(d["q_b_proj_weight"]),
                                     pk(d["kv_a_proj_weight"]), pk(d["w_kc"]), d["norm_weight1"], d["norm_weight2"], d["pos"],
                                     d["cos_sin_cache"], 1e-6, False, False, None, None, None, prepack, None)
    R = recipes.ABSORB_DIMS["kv_lora_rank"]
    assert q.shape == g["q"].shape and k.shape == g["k"].shape and v.shape == g["v"].shape
    assert close(g["q"][..., R:], q[..., R:]), "rotary slice of q"
    assert close(g["q"], q) and close(g["k"], k)
    assert ref_pred(g["v"], v)          # no cancellation on this path: the reference's predicate holds as is


@pytest.mark.parametrize("case", recipes.ABSORB_CASES, ids=lambda c: c[0])
def test_qkv_proj_with_rope_int8(ops, case):
    name, B, hidden, seed = case
    g, _ = load_golden("absorb_" + name)
    inp = recipes.absorb_inputs(B, hidden, seed)
    d = {k: v.cuda() for k, v in inp.items()}
    w = [quant_int8_rowwise(inp[n], floor=1e-7) for n in ("q_a_proj_weight", "q_b_proj_weight", "kv_a_proj_weight")]
    wq = [ops.convert_weight_packed(x[0].cuda()) for x in w]
    ws = [x[1].reshape(-1, 1).cuda() for x in w]
    q, k, v = ops.qkv_proj_with_rope(d["hidden_states"], wq[0], wq[1], wq[2], ops.convert_weight_packed(d["w_kc"]),
                                     d["norm_weight1"], d["norm_weight2"], d["pos"], d["cos_sin_cache"], 1e-6, True, False,
                                     ws[0], ws[1], ws[2], True, None)
    R = recipes.ABSORB_DIMS["kv_lora_rank"]
    # W8A8: q passes through two per-token quantisations whose round() amplifies any 1-ulp difference of its bf16 input
    # into a whole int8 step; measured 0.9-1.4 % relative RMS against the int8 oracle, where the reference's own bf16 and
    # int8 oracles differ by 2.1-2.3 % (k, v: one quantisation, 0.3 % vs 1.2-1.3 %).  Stated tolerance: 2 % / 0.6 %
    # relative RMS, elements within 8 / 3 bf16 ulps of the largest magnitude.
    assert within(g["q_int8"][..., R:], q[..., R:], 2e-2, 8)
    assert within(g["q_int8"], q, 2e-2, 8) and within(g["k_int8"], k, 6e-3, 3) and within(g["v_int8"], v, 6e-3, 3)


def test_qkv_proj_with_rope_cpu_tensors_and_errors(ops):
    """CPU tensors (what the reference script builds) are staged through the GPU; bad arguments raise."""
    name, B, hidden, seed = recipes.ABSORB_CASES[1]
    g, _ = load_golden("absorb_" + name)
    d = recipes.absorb_inputs(B, hidden, seed)
    q, k, v = ops.qkv_proj_with_rope(d["hidden_states"], d["q_a_proj_weight"], d["q_b_proj_weight"], d["kv_a_proj_weight"],
                                     d["w_kc"], d["norm_weight1"], d["norm_weight2"], d["pos"], d["cos_sin_cache"], 1e-6, False,
                                     False, None, None, None, False, None)
    assert q.device.type == "cpu" and close(g["q"], q) and close(g["k"], k) and ref_pred(g["v"], v)
    with pytest.raises(RuntimeError):
        ops.qkv_proj_with_rope(d["hidden_states"], d["q_a_proj_weight"], d["q_b_proj_weight"], d["kv_a_proj_weight"], d["w_kc"],
                               d["norm_weight1"], d["norm_weight2"], d["pos"], d["cos_sin_cache"], 1e-6, True, False, None, None,
                               None, False, None)   # int8 without scales
