"""GPU parity of extend_attention_cpu / decode_attention_cpu through torch.ops.sgl_kernel.
Pass predicates are the reference's: utils.compare on bf16 for extend (/root/reference/test_extend.py:188), cosine
similarity > 0.99 + allclose(atol=3e-2) + bit-exact KV-cache update for decode (/root/reference/test_mla.py:169-175)."""
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


def cuda(d):
    return {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


@pytest.mark.parametrize("case", recipes.EXTEND_CASES, ids=lambda c: c[0])
def test_extend_attention(ops, case):
    name, B, N_CTX, HQ, HKV, D, DV, mla, seed = case
    g, _ = load_golden("extend_" + name)
    inp = recipes.extend_inputs(B, N_CTX, HQ, HKV, D, DV, mla, seed)
    d = cuda(inp)
    o = torch.full((inp["q_extend"].shape[0], HQ, DV), float("nan"), dtype=torch.bfloat16, device="cuda")
    ret = ops.extend_attention_cpu(d["q_extend"], d["k_extend"], d["v_extend"], o, d["k_buffer"], d["v_buffer"],
                                   d["req_to_tokens"], d["b_req_idx"], d["b_seq_len"], d["b_extend"],
                                   d["b_start_loc_extend"], int(inp["b_extend"].max()), 1.0 / D ** 0.5, 0.0)
    assert ret is None
    assert torch.allclose(g["ref_out"], o.cpu(), rtol=1e-2, atol=1e-2), name
    ref32 = oattn.extend_attention(inp["q_extend"], inp["k_buffer"], inp["v_buffer"], inp["req_to_tokens"], inp["b_req_idx"],
                                   inp["b_seq_len"], inp["b_prefix"], inp["b_extend"], 1.0 / D ** 0.5)
    err = (o.float().cpu() - ref32).norm() / ref32.norm()
    assert err < 5e-3, f"{name}: relative RMS error {err:.2e}"


def test_extend_attention_logit_cap_and_int64_pages(ops):
    name, B, N_CTX, HQ, HKV, D, DV, mla, seed = recipes.EXTEND_CASES[0]
    inp = recipes.extend_inputs(B, N_CTX, HQ, HKV, D, DV, mla, seed)
    d = cuda(inp)
    o = torch.empty(inp["q_extend"].shape[0], HQ, DV, dtype=torch.bfloat16, device="cuda")
    ops.extend_attention_cpu(d["q_extend"] * 4, d["k_extend"], d["v_extend"], o, d["k_buffer"], d["v_buffer"],
                             d["req_to_tokens"].long(), d["b_req_idx"].int(), d["b_seq_len"].int(), d["b_extend"].long(),
                             d["b_start_loc_extend"].long(), int(inp["b_extend"].max()), 1.0 / D ** 0.5, 5.0)
    ref = oattn.extend_attention(inp["q_extend"] * 4, inp["k_buffer"], inp["v_buffer"], inp["req_to_tokens"], inp["b_req_idx"],
                                 inp["b_seq_len"], inp["b_prefix"], inp["b_extend"], 1.0 / D ** 0.5, logit_cap=5.0)
    assert torch.allclose(ref.bfloat16(), o.cpu(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("case", recipes.DECODE_CASES, ids=lambda c: c[0])
def test_decode_attention(ops, case):
    name, B, HQ, HKV, D, DV, seq_len, v_alias, seed = case
    g, _ = load_golden("decode_" + name)
    inp = recipes.decode_inputs(B, HQ, HKV, D, DV, seq_len, v_alias, seed)
    kb = inp["k_buffer"].cuda()
    key = inp["key"].cuda()
    if v_alias:          # exactly the reference's aliasing: v is a narrow view of k (test_mla.py:83,91)
        vb, value = kb.narrow(2, 0, DV), key.narrow(2, 0, DV)
    else:
        vb, value = inp["v_buffer"].cuda(), inp["value"].cuda()
    o = torch.zeros(B, HQ, DV, dtype=torch.bfloat16, device="cuda")
    logits = torch.empty(B, HQ, 8, DV + 1, dtype=torch.float32, device="cuda")
    ret = ops.decode_attention_cpu(inp["q"].cuda(), kb, vb, o, key, value, inp["loc"].cuda(), logits,
                                   inp["req_to_token"].cuda(), inp["b_req_idx"].cuda(), inp["b_seq_len"].cuda(),
                                   1.0 / D ** 0.5, 0.0)
    assert ret is None
    ref = g["ref_out"].float()
    cos = torch.nn.functional.cosine_similarity(o.float().cpu().flatten(), ref.flatten(), dim=0)
    assert cos > 0.99, f"{name}: cos_sim {cos}"
    assert torch.allclose(o.cpu().float(), ref, atol=3e-2), name
    # KV-cache update is bit-exact and touches nothing else
    expect_k = inp["k_buffer"].clone()
    expect_k[inp["loc"]] = inp["key"]
    assert torch.equal(kb.cpu(), expect_k)
    if not v_alias:
        expect_v = inp["v_buffer"].clone()
        expect_v[inp["loc"]] = inp["value"]
        assert torch.equal(vb.cpu(), expect_v)
    # tighter: against the fp32 oracle
    kb2 = inp["k_buffer"].clone()
    vb2 = kb2.narrow(2, 0, DV) if v_alias else inp["v_buffer"].clone()
    ref32 = oattn.decode_attention(inp["q"], kb2, vb2, inp["key"], inp["key"].narrow(2, 0, DV) if v_alias else inp["value"],
                                   inp["loc"], inp["req_to_token"], inp["b_req_idx"], inp["b_seq_len"], 1.0 / D ** 0.5)
    err = (o.float().cpu() - ref32).norm() / ref32.norm()
    assert err < 6e-3, f"{name}: relative RMS error {err:.2e}"


def test_decode_attention_ragged_lengths_and_int32_index(ops):
    """Different sequence lengths per request (some shorter than the split count), int32 page table and loc."""
    B, HQ, HKV, D, DV = 5, 16, 2, 128, 128
    lens = [1, 5, 64, 200, 777]
    L = max(lens)
    g = torch.Generator().manual_seed(12)
    total = B * L
    q = torch.randn(B, HQ, D, generator=g).bfloat16()
    kb = torch.randn(total, HKV, D, generator=g).bfloat16()
    vb = torch.randn(total, HKV, DV, generator=g).bfloat16()
    key = torch.randn(B, HKV, D, generator=g).bfloat16()
    value = torch.randn(B, HKV, DV, generator=g).bfloat16()
    perm = torch.randperm(total, generator=g)
    rtt = perm.view(B, L).to(torch.int32)
    seq = torch.tensor(lens)
    loc = torch.stack([rtt[b, lens[b] - 1] for b in range(B)]).to(torch.int32)   # the new token is the last position
    o = torch.zeros(B, HQ, DV, dtype=torch.bfloat16, device="cuda")
    logits = torch.empty(B, HQ, 8, DV + 1, dtype=torch.float32, device="cuda")
    kbd, vbd = kb.cuda(), vb.cuda()
    ops.decode_attention_cpu(q.cuda(), kbd, vbd, o, key.cuda(), value.cuda(), loc.cuda(), logits, rtt.cuda(),
                             torch.arange(B).cuda(), seq.cuda(), 1.0 / D ** 0.5, 0.0)
    kb2, vb2 = kb.clone(), vb.clone()
    ref = oattn.decode_attention(q, kb2, vb2, key, value, loc, rtt, torch.arange(B), seq, 1.0 / D ** 0.5)
    assert torch.equal(kbd.cpu(), kb2) and torch.equal(vbd.cpu(), vb2)
    assert torch.allclose(o.cpu().float(), ref, atol=3e-2)
    assert (o.float().cpu() - ref).norm() / ref.norm() < 6e-3


@pytest.mark.parametrize("shape", [(6, 22, 1, 576, 512, True), (7, 16, 4, 128, 128, False), (3, 8, 8, 64, 64, False)],
                         ids=["mla", "gqa", "mha64"])
def test_decode_attention_cache_write_inside_the_kernel(ops, knob, shape):
    """The cache write rides in the attention kernel for small batches (DecodeParams::fold_write).  The reference writes ALL new
    rows first and then attends (/root/reference/test_mla.py:27-28), and its own inputs draw `loc` anywhere in the pool (:96),
    so a request's page list may name ANOTHER request's new row: here request b reads the slot request b+1 writes, at a known
    position.  Checked against the oracle, and bit for bit against the separate-launch form (SGLK_DEC_FOLD=0) -- outputs, both
    caches and the partial logits -- and with a `key` whose rows are not 16-byte aligned (falls back to the separate launch)."""
    B, HQ, HKV, D, DV, alias = shape
    L = 150
    g = torch.Generator().manual_seed(77)
    total = B * L + B
    q = torch.randn(B, HQ, D, generator=g).bfloat16()
    kb = torch.randn(total, HKV, D, generator=g).bfloat16()
    vb = None if alias else torch.randn(total, HKV, DV, generator=g).bfloat16()
    key = (torch.randn(B, HKV, D, generator=g) * 3).bfloat16()          # loud new rows: a stale read shows
    value = None if alias else (torch.randn(B, HKV, DV, generator=g) * 3).bfloat16()
    rtt = torch.randperm(B * L, generator=g).view(B, L)
    loc = B * L + torch.arange(B)                                        # fresh slots ...
    lens = torch.tensor([L - 7 * b for b in range(B)])
    for b in range(B):
        rtt[b, lens[b] - 1] = loc[b]                                     # ... the request's own new token last,
        rtt[b, 3 + b] = loc[(b + 1) % B]                                 # and another request's new row in the middle
    scale = 1.0 / D ** 0.5

    def run(key_t, value_t):
        kbd = kb.cuda()
        vbd = kbd.narrow(2, 0, DV) if alias else vb.cuda()
        o = torch.zeros(B, HQ, DV, dtype=torch.bfloat16, device="cuda")
        logits = torch.zeros(B, HQ, 8, DV + 1, dtype=torch.float32, device="cuda")
        ops.decode_attention_cpu(q.cuda(), kbd, vbd, o, key_t, key_t.narrow(2, 0, DV) if alias else value_t, loc.cuda(), logits,
                                 rtt.cuda(), torch.arange(B).cuda(), lens.cuda(), scale, 0.0)
        return o.cpu(), kbd.cpu(), vbd.cpu(), logits.cpu()

    keyd, valued = key.cuda(), None if alias else value.cuda()
    folded = run(keyd, valued)
    knob(SGLK_DEC_FOLD=0)
    separate = run(keyd, valued)
    knob(SGLK_DEC_FOLD=None)
    for a, b, what in zip(folded, separate, ("output", "k_buffer", "v_buffer", "partial logits")):
        assert torch.equal(a, b), f"folded and separate cache write differ in the {what}"
    # rows that start 2 bytes off a 16-byte boundary: the vector path must not be taken
    pad = torch.zeros(B, HKV, D + 1, dtype=torch.bfloat16, device="cuda")
    pad[:, :, 1:] = keyd
    off = run(pad[:, :, 1:], valued)
    assert torch.equal(off[0], folded[0]) and torch.equal(off[1], folded[1])
    kb2 = kb.clone()
    vb2 = kb2.narrow(2, 0, DV) if alias else vb.clone()
    ref = oattn.decode_attention(q, kb2, vb2, key, key.narrow(2, 0, DV) if alias else value, loc, rtt, torch.arange(B), lens, scale)
    assert torch.equal(folded[1], kb2) and torch.equal(folded[2], vb2), "cache contents after the call"
    err = (folded[0].float() - ref).norm() / ref.norm()
    assert err < 6e-3, f"relative RMS error {err:.2e}: a stale cache row would be far off"


@pytest.mark.parametrize("shape", [(5, 22, 1, 576, 512, True), (9, 32, 4, 128, 128, False)], ids=["mla", "gqa"])
def test_decode_attention_one_split_writes_the_output_itself(ops, knob, shape):
    """With one split per request (large batches; forced here by SGLK_DEC_SPLITS=1) the attention kernel stores the rounded output and
    no merge kernel runs.  Same bits as the merge of one split: checked against the same call into an output whose rows are not 8-byte
    aligned (that one goes through the merge), and against the oracle."""
    B, HQ, HKV, D, DV, alias = shape
    L = 333
    g = torch.Generator().manual_seed(5)
    total = B * L
    q = torch.randn(B, HQ, D, generator=g).bfloat16()
    kb = torch.randn(total, HKV, D, generator=g).bfloat16()
    vb = None if alias else torch.randn(total, HKV, DV, generator=g).bfloat16()
    key = torch.randn(B, HKV, D, generator=g).bfloat16()
    value = None if alias else torch.randn(B, HKV, DV, generator=g).bfloat16()
    rtt = torch.randperm(total, generator=g).view(B, L)
    lens = torch.tensor([L - 11 * b for b in range(B)])
    loc = torch.stack([rtt[b, lens[b] - 1] for b in range(B)])
    scale = 1.0 / D ** 0.5
    knob(SGLK_DEC_SPLITS=1)

    def run(o):
        kbd = kb.cuda()
        vbd = kbd.narrow(2, 0, DV) if alias else vb.cuda()
        logits = torch.zeros(B, HQ, 8, DV + 1, dtype=torch.float32, device="cuda")
        kd = key.cuda()
        ops.decode_attention_cpu(q.cuda(), kbd, vbd, o, kd, kd.narrow(2, 0, DV) if alias else value.cuda(), loc.cuda(), logits,
                                 rtt.cuda(), torch.arange(B).cuda(), lens.cuda(), scale, 0.0)
        return o.cpu(), logits.cpu()

    direct, lg_direct = run(torch.zeros(B, HQ, DV, dtype=torch.bfloat16, device="cuda"))
    wide = torch.zeros(B, HQ, DV + 2, dtype=torch.bfloat16, device="cuda")
    merged, lg_merged = run(wide[:, :, 1:DV + 1])
    assert lg_direct.abs().sum() == 0 and lg_merged.abs().sum() > 0, "the direct form must not touch attn_logits; the other one must"
    assert torch.equal(direct, merged)
    kb2 = kb.clone()
    vb2 = kb2.narrow(2, 0, DV) if alias else vb.clone()
    ref = oattn.decode_attention(q, kb2, vb2, key, key.narrow(2, 0, DV) if alias else value, loc, rtt, torch.arange(B), lens, scale)
    assert (direct.float() - ref).norm() / ref.norm() < 6e-3


# ---- BASELINE.json config 3 at its own sizes (seqlen <= 8k): /root/reference/bench_extend.py:107-112, test_mla.py:178-186 ----
@pytest.mark.parametrize("case", recipes.EXTEND_BIG_CASES, ids=lambda c: c[0])
def test_extend_attention_bench_sizes(ops, case):
    """The shapes the reference benches (ctx 4096 x 32/4 heads, 8192 x 16/2, MLA-like prefill 4 x 3500 x 22 heads, D=192):
    every row computed on the GPU; the golden holds the reference oracle's (_run_sdpa_forward_extend) output for ~160
    sampled token rows x all heads -- both ends of the sequence, block edges, random rows in between."""
    name, B, N_CTX, HQ, HKV, D, DV, mla, seed = case
    g, meta = load_golden("extend_" + name)
    inp = recipes.extend_inputs_fixed(B, N_CTX, HQ, HKV, D, DV, mla, seed)
    T = inp["q_extend"].shape[0]
    assert T == int(meta["tokens"])
    d = cuda(inp)
    o = torch.full((T, HQ, DV), float("nan"), dtype=torch.bfloat16, device="cuda")
    ret = ops.extend_attention_cpu(d["q_extend"], d["k_extend"], d["v_extend"], o, d["k_buffer"], d["v_buffer"],
                                   d["req_to_tokens"], d["b_req_idx"], d["b_seq_len"], d["b_extend"],
                                   d["b_start_loc_extend"], int(inp["b_extend"].max()), 1.0 / D ** 0.5, 0.0)
    assert ret is None
    assert torch.isfinite(o.float()).all(), "a row was not written"
    rows = g["rows"]
    got = o[rows.cuda()].cpu()
    assert torch.allclose(g["ref_out_rows"], got, rtol=1e-2, atol=1e-2), name        # utils.compare, test_extend.py:188
    ref = g["ref_out_rows"].float()
    err = (got.float() - ref).norm() / ref.norm()
    assert err < 8e-3, f"{name}: relative RMS error {err:.2e} against the (bf16-rounded) reference output"


@pytest.mark.parametrize("case", recipes.DECODE_BIG_CASES, ids=lambda c: c[0])
def test_decode_attention_seq4096(ops, case):
    """Decode over 4096 cached keys: MLA B=40 (test_mla.py:178-186 at the config's length) and GQA B=64; reference predicates
    (cosine similarity > 0.99, allclose(atol=3e-2), bit-exact cache write) against the reference oracle's whole output."""
    name, B, HQ, HKV, D, DV, seq_len, v_alias, seed = case
    g, _ = load_golden("decode_" + name)
    inp = recipes.decode_inputs(B, HQ, HKV, D, DV, seq_len, v_alias, seed)
    kb = inp["k_buffer"].cuda()
    key = inp["key"].cuda()
    if v_alias:
        vb, value = kb.narrow(2, 0, DV), key.narrow(2, 0, DV)
    else:
        vb, value = inp["v_buffer"].cuda(), inp["value"].cuda()
    o = torch.zeros(B, HQ, DV, dtype=torch.bfloat16, device="cuda")
    logits = torch.empty(B, HQ, 8, DV + 1, dtype=torch.float32, device="cuda")
    ops.decode_attention_cpu(inp["q"].cuda(), kb, vb, o, key, value, inp["loc"].cuda(), logits,
                             inp["req_to_token"].cuda(), inp["b_req_idx"].cuda(), inp["b_seq_len"].cuda(), 1.0 / D ** 0.5, 0.0)
    ref = g["ref_out"].float()
    cos = torch.nn.functional.cosine_similarity(o.float().cpu().flatten(), ref.flatten(), dim=0)
    assert cos > 0.99, f"{name}: cos_sim {cos}"
    assert torch.allclose(o.cpu().float(), ref, atol=3e-2), name
    err = (o.float().cpu() - ref).norm() / ref.norm()
    assert err < 1e-2, f"{name}: relative RMS error {err:.2e} against the (bf16-rounded) reference output"
    rows = inp["loc"]
    assert torch.equal(kb[rows.cuda()].cpu(), inp["key"]), "KV-cache write"
    untouched = torch.ones(kb.shape[0], dtype=torch.bool)
    untouched[rows] = False
    sel = torch.nonzero(untouched).flatten()[:: max(1, kb.shape[0] // 4096)]
    assert torch.equal(kb[sel.cuda()].cpu(), inp["k_buffer"][sel]), "rows other than `loc` must stay as they were"
