"""Pins the oracle (torch + plain-C restatements) against golden vectors produced by the
reference's own embedded oracles (tests/golden/make_golden.py).  CPU only."""
import hashlib

import pytest
import torch

import recipes
from conftest import load_golden
from oracle import c_oracle, moe


def _sha(*ts):
    h = hashlib.sha256()
    for t in ts:
        t = t.contiguous()
        if t.dtype == torch.float8_e4m3fn:
            t = t.view(torch.uint8)
        elif t.dtype == torch.bfloat16:
            t = t.view(torch.int16)
        h.update(t.numpy().tobytes())
    return h.hexdigest()


def fp8_case_inputs(case):
    name, M, N, K, E, topk, bn, bk, masked, seed, full = case
    g, meta = load_golden("moe_fp8_" + name)
    inp = recipes.moe_fp8_inputs(M, N, K, E, topk, bn, bk, masked, seed)
    assert _sha(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], inp["topk_weight"], inp["topk_ids"]) \
        == meta["input_sha256"], "input recipe drifted from the golden file"
    if full:  # stored inputs must equal the regenerated ones bit for bit
        for k in ("a", "w1", "w2", "w1s", "w2s", "topk_weight", "topk_ids"):
            x, y = g[k], inp[k]
            if x.dtype == torch.float8_e4m3fn:
                x, y = x.view(torch.uint8), y.view(torch.uint8)
            assert torch.equal(x, y), k
    return inp, g["ref_out_f32"], (bn, bk)


def _check_fp32_or_bf16(out, ref, masked):
    if masked:
        # /root/reference/test_moe_offloading_cpu.py:52 rounds its oracle's result to bf16 (.to(old_dtype)):
        # equal after the same rounding, up to one bf16 ulp where the fp32 sums differ in the last bits
        torch.testing.assert_close(out.bfloat16().float(), ref, rtol=2 ** -7, atol=1e-6)
    else:
        # same fp32 math, different summation grouping only
        torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("case", recipes.MOE_FP8_CASES, ids=lambda c: c[0])
def test_torch_oracle_fp8_matches_reference(case):
    inp, ref, block = fp8_case_inputs(case)
    out = moe.fused_experts_fp8(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], block,
                                inp["topk_weight"], inp["topk_ids"])
    _check_fp32_or_bf16(out, ref, case[8])


@pytest.mark.parametrize("case", recipes.MOE_FP8_CASES, ids=lambda c: c[0])
def test_c_oracle_fp8_matches_reference(case):
    inp, ref, block = fp8_case_inputs(case)
    out = c_oracle.fused_experts_fp8(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], block,
                                     inp["topk_weight"], inp["topk_ids"])
    _check_fp32_or_bf16(out, ref, case[8])
    # and it passes the reference's own predicate (utils.compare, bf16 tolerance)
    assert moe.allclose_ref(ref.bfloat16(), out.bfloat16())


@pytest.mark.parametrize("case", recipes.MOE_INT8_CASES, ids=lambda c: c[0])
def test_torch_oracle_int8_matches_reference(case):
    name, M, N, K, E, topk, seed, full = case
    g, meta = load_golden("moe_int8_" + name)
    inp = recipes.moe_int8_inputs(M, N, K, E, topk, seed)
    assert _sha(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], inp["topk_weight"], inp["topk_ids"]) \
        == meta["input_sha256"]
    out = moe.fused_experts_int8(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"],
                                 inp["topk_weight"], inp["topk_ids"])
    ref = g["ref_out"].float()          # reference returns bf16 (.to(a.dtype))
    torch.testing.assert_close(out.bfloat16().float(), ref, rtol=1e-2, atol=1e-4)


@pytest.mark.parametrize("case", recipes.MOE_BF16_CASES, ids=lambda c: c[0])
def test_torch_oracle_bf16_matches_reference(case):
    name, M, N, K, E, topk, renorm, seed, full = case
    g, meta = load_golden("moe_bf16_" + name)
    inp = recipes.moe_bf16_inputs(M, N, K, E, topk, seed)
    assert _sha(inp["a"], inp["w1"], inp["w2"], inp["score"]) == meta["input_sha256"]
    w, ids = moe.softmax_topk(inp["score"], topk, renorm)
    out = moe.fused_experts_f32(inp["a"], inp["w1"].float(), inp["w2"].float(), w, ids)
    # the reference's bf16 oracle (test_moe.py:22-54) computes in bf16 end to end; ours in fp32:
    # they agree to the reference's own bf16 tolerance
    assert moe.allclose_ref(g["ref_out"], out.bfloat16())


# ---- dense GEMMs and shared expert ---------------------------------------------------------------------------------
from oracle import gemm as ogemm  # noqa: E402


@pytest.mark.parametrize("case", recipes.GEMM_INT8_CASES, ids=lambda c: c[0])
def test_oracle_gemm_int8(case):
    name, M, N, K, has_bias, seed = case
    g, meta = load_golden("gemm_int8_" + name)
    inp = recipes.gemm_int8_inputs(M, N, K, has_bias, seed)
    assert _sha(inp["A"], inp["Bq"], inp["Bs"]) == meta["input_sha256"]
    q, s = ogemm.per_token_quant_int8(inp["A"])
    assert torch.equal(q, g["ref_Aq"]) and torch.equal(s, g["ref_As"])          # integer work: bit-exact
    out = ogemm.int8_scaled_mm(q, s, inp["Bq"], inp["Bs"], inp.get("bias"))
    assert moe.allclose_ref(g["ref_out"], out.bfloat16())


@pytest.mark.parametrize("case", recipes.GEMM_FP8_CASES, ids=lambda c: c[0])
def test_oracle_gemm_fp8(case):
    name, M, N, K, has_bias, chunk, seed = case
    g, meta = load_golden("gemm_fp8_" + name)
    inp = recipes.gemm_fp8_inputs(M, N, K, has_bias, chunk, seed)
    assert _sha(inp["data"].contiguous(), inp["w"], inp["scales"]) == meta["input_sha256"]
    out = ogemm.fp8_scaled_mm(inp["data"], inp["w"], inp["scales"], (64, 128), inp.get("bias"))
    torch.testing.assert_close(out, g["ref_out_f32"], rtol=1e-4, atol=1e-5)
    assert moe.allclose_ref(g["ref_out_bf16"], out.bfloat16())                  # the reference's own bf16 oracle


@pytest.mark.parametrize("case", recipes.GEMM_BF16_CASES, ids=lambda c: c[0])
def test_oracle_gemm_bf16(case):
    name, M, N, K, has_bias, seed = case
    g, meta = load_golden("gemm_bf16_" + name)
    inp = recipes.gemm_bf16_inputs(M, N, K, has_bias, seed)
    assert _sha(inp["mat1"], inp["mat2"]) == meta["input_sha256"]
    out = ogemm.linear_bf16(inp["mat1"], inp["mat2"], inp.get("bias"))
    assert torch.equal(out.bfloat16(), g["ref_out"])


@pytest.mark.parametrize("case", recipes.SHARED_CASES, ids=lambda c: c[0])
def test_oracle_shared_expert(case):
    name, m, n, k, rsf, seed = case
    g, meta = load_golden("shared_" + name)
    inp = recipes.shared_inputs(m, n, k, seed)
    assert _sha(inp["hs"], inp["w1"], inp["w2"], inp["fused"]) == meta["input_sha256"]
    out = moe.shared_expert_f32(inp["hs"], inp["w1"], inp["w2"], inp["fused"], rsf)
    assert moe.allclose_ref(g["ref_bf16"], out.bfloat16())
    w1q, w1s = moe.quant_int8_rowwise(inp["w1"])
    assert torch.equal(w1q, g["w1q"])
    out8 = moe.shared_expert_int8(inp["hs"], g["w1q"], g["w2q"], g["w1s"], g["w2s"], inp["fused"], rsf)
    assert moe.allclose_ref(g["ref_int8"], out8.bfloat16())


@pytest.mark.parametrize("case", recipes.SHARED_FP8_CASES, ids=lambda c: c[0])
def test_oracle_shared_expert_fp8(case):
    name, M, N, K, rsf, seed = case
    g, meta = load_golden("shared_fp8_" + name)
    inp = recipes.shared_fp8_inputs(M, N, K, seed)
    assert _sha(inp["a"], inp["w1"], inp["w2"], inp["w1s"], inp["w2s"], inp["fused"]) == meta["input_sha256"]
    w1 = moe.dequant_block_fp8(inp["w1"], inp["w1s"], 64, 128)
    w2 = moe.dequant_block_fp8(inp["w2"], inp["w2s"], 64, 128)
    out = moe.shared_expert_f32(inp["a"], w1, w2, inp["fused"], rsf)
    torch.testing.assert_close(out, g["ref_out_f32"], rtol=1e-4, atol=1e-5)


# ---- routing and row kernels ------------------------------------------------------------------------------------------
from oracle import elementwise as oew  # noqa: E402
from oracle import routing  # noqa: E402


def routing_keys(g, biased, G, topk_group):
    """Selection key of every expert (choice inside the selected groups, -inf / -1 outside) and the unbiased scores."""
    gating = g["gating"].float()
    M, E = gating.shape
    if biased:
        scores = gating.sigmoid()
        choice = scores + g["bias"].float().unsqueeze(0)
        gs = torch.sort(choice.view(M, G, -1), dim=-1, descending=True).values[..., :2].sum(-1)
    else:
        scores = torch.softmax(gating, dim=-1)
        choice = scores
        gs = scores.view(M, G, -1).max(dim=-1).values
    return scores, choice, gs


def check_routing(w, ids, g, biased, G, topk_group, topk, tag=""):
    """Tie-aware equivalence with the reference oracle's (ref_w, ref_ids): the reference compares scattered [M,E]
    matrices (/root/reference/test_grouped_topk.py:71-75), which is only well defined without ties at the cut, so:
    the multiset of selection keys must be identical, the multiset of weights equal, and where the k-th and
    (k+1)-th keys differ the scattered matrices must agree too."""
    scores, choice, gs = routing_keys(g, biased, G, topk_group)
    M, E = scores.shape
    ref_ids, ref_w = g["ref_ids"].long(), g["ref_w"]
    ids = ids.long().cpu()
    w = w.cpu()
    for m in range(M):
        assert len(set(ids[m].tolist())) == topk, f"{tag} row {m}: duplicate ids"
    if biased:   # every pick is pinned by its key; the softmax variant's zero-weight picks are not (masked_fill(0.0))
        mine_k = torch.sort(choice.gather(1, ids), dim=1).values
        ref_k = torch.sort(choice.gather(1, ref_ids), dim=1).values
        assert torch.equal(mine_k, ref_k), f"{tag}: selected keys differ from the reference"
    res = torch.zeros(M, E).scatter_(1, ids, w)
    ref = torch.zeros(M, E).scatter_(1, ref_ids, ref_w)
    row_ok = torch.isclose(res, ref, rtol=1e-5, atol=1e-5).all(dim=1)
    # rows that disagree must be explained by a tie among the candidates' keys
    for m in (~row_ok).nonzero().flatten().tolist():
        a, b = set(ids[m].tolist()), set(ref_ids[m].tolist())
        ka = sorted(choice[m, list(a - b)].tolist())
        kb = sorted(choice[m, list(b - a)].tolist())
        assert ka == kb, f"{tag} row {m}: differs from the reference beyond a tie ({ka} vs {kb})"
    return int(row_ok.sum())


@pytest.mark.parametrize("case", recipes.TOPK_CASES, ids=lambda c: c[0])
def test_oracle_routing_matches_reference(case):
    name, M, E, G, topk, topk_group, renorm, biased, seed = case
    g, _ = load_golden("topk_" + name)
    if biased:
        w, ids = routing.biased_grouped_topk(g["gating"], g["bias"], topk, renorm, G, topk_group)
    else:
        w, ids = routing.grouped_topk(g["gating"], topk, renorm, G, topk_group)
    exact_rows = check_routing(w, ids, g, biased, G, topk_group, topk, name)
    assert exact_rows >= int(0.9 * M)      # ties are the exception


@pytest.mark.parametrize("case", recipes.NORM_CASES, ids=lambda c: c[0])
def test_oracle_rmsnorm(case):
    name, rows, hidden, dtype, seed = case
    g, _ = load_golden("norm_" + name)
    inp = recipes.norm_inputs(rows, hidden, dtype, seed)
    assert torch.equal(oew.rmsnorm(inp["x"], inp["w"]), g["ref_out"])
    o, r = oew.rmsnorm(inp["x"], inp["w"], 1e-6, inp["res"])
    assert torch.equal(o, g["ref_fused_out"]) and torch.equal(r, g["ref_fused_res"])


@pytest.mark.parametrize("case", recipes.ACT_CASES, ids=lambda c: c[0])
def test_oracle_silu_and_mul(case):
    name, rows, two_d, dtype, seed = case
    g, _ = load_golden("act_" + name)
    assert torch.equal(oew.silu_and_mul(recipes.act_inputs(rows, two_d, dtype, seed)["x"]), g["ref_out"])


# ---- attention -----------------------------------------------------------------------------------------------------------
from oracle import attention as oattn  # noqa: E402


@pytest.mark.parametrize("case", recipes.EXTEND_CASES[:4], ids=lambda c: c[0])
def test_oracle_extend_attention(case):
    name, B, N_CTX, HQ, HKV, D, DV, mla, seed = case
    g, _ = load_golden("extend_" + name)
    inp = recipes.extend_inputs(B, N_CTX, HQ, HKV, D, DV, mla, seed)
    out = oattn.extend_attention(inp["q_extend"], inp["k_buffer"], inp["v_buffer"], inp["req_to_tokens"], inp["b_req_idx"],
                                 inp["b_seq_len"], inp["b_prefix"], inp["b_extend"], 1.0 / D ** 0.5)
    assert moe.allclose_ref(g["ref_out"], out.bfloat16())


@pytest.mark.parametrize("case", [recipes.DECODE_CASES[0], recipes.DECODE_CASES[4]], ids=lambda c: c[0])
def test_oracle_decode_attention(case):
    name, B, HQ, HKV, D, DV, seq_len, v_alias, seed = case
    g, _ = load_golden("decode_" + name)
    inp = recipes.decode_inputs(B, HQ, HKV, D, DV, seq_len, v_alias, seed)
    kb = inp["k_buffer"].clone()
    vb = kb.narrow(2, 0, DV) if v_alias else inp["v_buffer"].clone()
    value = inp["key"].narrow(2, 0, DV) if v_alias else inp["value"]
    out = oattn.decode_attention(inp["q"], kb, vb, inp["key"], value, inp["loc"], inp["req_to_token"], inp["b_req_idx"],
                                 inp["b_seq_len"], 1.0 / D ** 0.5)
    assert torch.allclose(out.bfloat16().float(), g["ref_out"].float(), atol=3e-2)      # test_mla.py:173
    assert torch.equal(kb[inp["loc"]], inp["key"])


@pytest.mark.parametrize("case", recipes.VARLEN_CASES, ids=lambda c: c[0])
def test_oracle_flash_attn_varlen(case):
    name, batch, max_q, max_k, H, Hkv, D, DV, causal, varlen, seed = case
    g, _ = load_golden("varlen_" + name)
    inp = recipes.varlen_inputs(batch, max_q, max_k, H, Hkv, D, DV, varlen, seed)
    out = oattn.flash_attn_varlen(inp["q"], inp["k"], inp["v"], inp["cu_q"], inp["cu_k"], causal)
    assert moe.allclose_ref(g["ref_out"], out.bfloat16())


@pytest.mark.parametrize("case", recipes.BMM_CASES, ids=lambda c: c[0])
def test_oracle_bmm(case):
    from oracle import gemm as ogemm
    name, B, M, N, K, chunk, seed = case
    g, _ = load_golden("bmm_" + name)
    inp = recipes.bmm_inputs(B, M, N, K, chunk, seed)
    assert moe.allclose_ref(g["ref_out"], ogemm.bmm(inp["mat1"], inp["mat2"]).bfloat16())


@pytest.mark.parametrize("case", recipes.MXFP4_CASES, ids=lambda c: c[0])
def test_oracle_mxfp4(case):
    """oracle/gemm.py's MX-fp4 quantiser, dequantiser and GEMM against the reference's MXFP4QuantizeUtil outputs."""
    from oracle import gemm as ogemm
    name, M, N, K, kind, has_bias, seed = case
    g, _ = load_golden("mxfp4_" + name)
    inp = recipes.mxfp4_inputs(M, N, K, kind, has_bias, seed, ogemm.mxfp4_quantize)
    assert torch.equal(inp["wq"], g["wq"]) and torch.equal(inp["ws"], g["ws"])          # quantiser, bit-exact
    assert torch.equal(ogemm.mxfp4_dequant(inp["wq"], inp["ws"]).bfloat16(), g["dq"])    # dequantiser, bit-exact
    out = ogemm.mxfp4_scaled_mm(inp["a"], inp["wq"], inp["ws"], inp["bias"])
    assert moe.allclose_ref(g["ref_out"], out.bfloat16())


# ---- qkv_proj_with_rope (oracle/absorb.py vs the reference's native_torch / native_torch_int8) --------------------------
@pytest.mark.parametrize("case", recipes.ABSORB_CASES, ids=lambda c: c[0])
def test_absorb_oracle_matches_reference_oracle(case):
    from oracle import absorb
    from oracle.gemm import quant_int8_rowwise
    name, B, hidden, seed = case
    g, _ = load_golden("absorb_" + name)
    inp = recipes.absorb_inputs(B, hidden, seed)
    common = (inp["w_kc"], inp["norm_weight1"], inp["norm_weight2"], inp["pos"], inp["cos_sin_cache"])
    q, k, v = absorb.qkv_proj_with_rope(inp["hidden_states"], inp["q_a_proj_weight"], inp["q_b_proj_weight"],
                                        inp["kv_a_proj_weight"], *common)
    assert torch.equal(q, g["q"]) and torch.equal(k, g["k"]) and torch.equal(v, g["v"])
    w = [quant_int8_rowwise(inp[n], floor=1e-7) for n in ("q_a_proj_weight", "q_b_proj_weight", "kv_a_proj_weight")]
    q8, k8, v8 = absorb.qkv_proj_with_rope(inp["hidden_states"], w[0][0], w[1][0], w[2][0], *common,
                                           scales=(w[0][1], w[1][1], w[2][1]))
    assert torch.equal(q8, g["q_int8"]) and torch.equal(k8, g["k_int8"]) and torch.equal(v8, g["v_int8"])
