"""GPU parity of the row kernels and the routing kernels through torch.ops.sgl_kernel:
silu_and_mul (/root/reference/test_activation.py), rmsnorm / fused_add_rmsnorm (/root/reference/test_norm.py),
grouped_topk / biased_grouped_topk (/root/reference/test_grouped_topk.py, test_biased_grouped_topk.py)."""
import pytest
import torch

import recipes
from conftest import load_golden
from oracle import routing
from test_oracle_golden import check_routing

pytestmark = pytest.mark.gpu

PRES = {torch.bfloat16: 1e-2, torch.float16: 1e-3}   # /root/reference/utils.py:3-7


@pytest.fixture(scope="module")
def ops():
    import sgl_kernel  # noqa: F401
    assert torch.cuda.is_available()
    return torch.ops.sgl_kernel


def ref_compare(out, ref):
    tol = PRES[ref.dtype]
    return torch.allclose(out.cpu(), ref, rtol=tol, atol=tol)


def ulp_close(out, ref, frac=0.999):
    """The roundings are placed where torch places them, so results match bit for bit except where the hardware
    exp / rsqrt differ from libm in the last bit before a rounding: >= 99.9 % equal, the rest within one ulp."""
    out = out.cpu()
    same = (out == ref).float().mean().item()
    a, b = out.view(torch.int16).int(), ref.view(torch.int16).int()
    # neighbouring representable values differ by 1 in the sign-magnitude bit pattern (same sign), or are +-tiny
    mag = lambda v: torch.where(v < 0, -(v & 0x7fff), v)   # noqa: E731  monotone integer key
    # (a last-bit difference in exp/rsqrt can flip the FIRST of the two roundings; times the weight that is <= 2 ulp)
    one_ulp = ((mag(a) - mag(b)).abs() <= 2).all().item()
    return same >= frac and one_ulp, same


@pytest.mark.parametrize("case", recipes.ACT_CASES, ids=lambda c: c[0])
def test_silu_and_mul(ops, case):
    from sgl_kernel.ops._kernels import silu_and_mul_cpu as legacy
    name, rows, two_d, dtype, seed = case
    g, _ = load_golden("act_" + name)
    x = recipes.act_inputs(rows, two_d, dtype, seed)["x"].cuda()
    out = ops.silu_and_mul_cpu(x)                                    # returning form (bench_silu_and_mul.py:31)
    out2 = torch.empty(rows, two_d // 2, dtype=dtype, device="cuda")
    assert legacy(out2, x) is None                                   # out-param form (test_activation.py:25)
    assert torch.equal(out, out2)
    assert ref_compare(out, g["ref_out"])
    ok, same = ulp_close(out, g["ref_out"], 0.99)
    assert ok, f"{name}: only {same:.4%} bit-identical"
    x3 = x.view(rows, 1, two_d).expand(rows, 1, two_d).contiguous()
    assert torch.equal(ops.silu_and_mul_cpu(x3).view(rows, -1), out)  # >2-D input


@pytest.mark.parametrize("case", recipes.NORM_CASES, ids=lambda c: c[0])
def test_rmsnorm_and_fused_add(ops, case):
    name, rows, hidden, dtype, seed = case
    g, _ = load_golden("norm_" + name)
    inp = {k: v.cuda() for k, v in recipes.norm_inputs(rows, hidden, dtype, seed).items()}
    out = torch.empty_like(inp["x"])
    assert ops.rmsnorm_cpu(out, inp["x"], inp["w"], 1e-6) is None     # test_norm.py:43-44
    assert ref_compare(out, g["ref_out"])
    ok, same = ulp_close(out, g["ref_out"])
    assert ok, f"{name}: only {same:.4%} bit-identical"
    x, res = inp["x"].clone(), inp["res"].clone()
    assert ops.fused_add_rmsnorm_cpu(x, res, inp["w"], 1e-6) is None  # in place on both (test_norm.py:56)
    assert torch.equal(res.cpu(), g["ref_fused_res"]), "residual = round(x + residual) must be bit-exact"
    assert ref_compare(x, g["ref_fused_out"])
    ok, same = ulp_close(x, g["ref_fused_out"])
    assert ok, f"{name}: fused only {same:.4%} bit-identical"


@pytest.mark.parametrize("case", recipes.TOPK_CASES, ids=lambda c: c[0])
def test_grouped_topk(ops, case):
    name, M, E, G, topk, topk_group, renorm, biased, seed = case
    g, _ = load_golden("topk_" + name)
    hidden, gating = g["hidden"].cuda(), g["gating"].cuda()
    w = torch.empty(M, topk, dtype=torch.float32, device="cuda")
    ids = torch.empty(M, topk, dtype=torch.int32, device="cuda")
    if biased:   # out-param 9-arg form, test_biased_grouped_topk.py:71-80
        assert ops.biased_grouped_topk_cpu(w, ids, hidden, gating, g["bias"].cuda(), topk, renorm, G, topk_group) is None
        ow, oids = routing.biased_grouped_topk(g["gating"], g["bias"], topk, renorm, G, topk_group)
    else:        # out-param 8-arg form, test_grouped_topk.py:61-69
        assert ops.grouped_topk_cpu(w, ids, hidden, gating, topk, renorm, G, topk_group) is None
        ow, oids = routing.grouped_topk(g["gating"], topk, renorm, G, topk_group)
        w2, ids2 = ops.grouped_topk_cpu(hidden, gating, topk, renorm, G, topk_group, 0, None, None)   # test_moe.py:61-70
        assert torch.equal(w2, w) and torch.equal(ids2, ids)
    # 1. equivalent to the reference's oracle up to ties
    check_routing(w, ids, g, biased, G, topk_group, topk, name)
    # 2. against this repo's oracle (same tie rule), ids BIT-EXACT:
    same_rows = (ids.cpu() == oids).all(dim=1)
    if not biased:
        # softmax variant: groups and experts are ranked by the logits (exact comparisons), no exp() rounding involved
        assert same_rows.all(), f"{name}: ids differ from the oracle in {(~same_rows).sum().item()} rows"
    else:
        # sigmoid + bias: the ranking key is a rounded quantity (the GPU's exp and torch's differ in the last ulps).  Every
        # row whose ids differ must be a GENUINE near-tie: nudging the key of the experts the kernel picked up by 8 ulps
        # (+ 1e-37 for flushed denormals) must make the oracle pick exactly the kernel's set, groups included.
        bad = torch.nonzero(~same_rows).flatten().tolist()
        gat, bia = g["gating"].float(), g["bias"].float()
        for m in bad:
            picked = torch.zeros(E, dtype=torch.bool)
            picked[ids[m].cpu().long()] = True
            key = gat[m].sigmoid() + bia
            nudge = torch.where(picked, key.abs() * (8 * 2.0 ** -23) + 1e-37, torch.zeros(E))
            _, ids_n = routing.biased_grouped_topk(gat[m:m + 1], bia, topk, renorm, G, topk_group, key_nudge=nudge.unsqueeze(0))
            assert set(ids_n[0].tolist()) == set(ids[m].cpu().tolist()), \
                f"{name}: row {m} differs from the oracle and is not a near-tie: kernel {sorted(ids[m].cpu().tolist())} " \
                f"oracle {sorted(oids[m].tolist())}"
        print(f"[topk] {name}: {len(bad)} of {M} rows differ from the oracle, all of them near-ties of sigmoid(x) + bias within 8 ulps")
    assert torch.allclose(w.cpu()[same_rows], ow[same_rows], rtol=2e-5, atol=1e-6)


def test_grouped_topk_exact_ids_on_separated_scores(ops):
    """With well separated logits (no near-ties) routing ids must equal the oracle's bit for bit — Qwen3 routing
    (128 experts, top-8, one group), f32 / bf16 / f16 gating."""
    M, E, topk = 777, 128, 8
    g = torch.Generator().manual_seed(3)
    base = torch.stack([torch.randperm(E, generator=g) for _ in range(M)]).float() * 0.37 - 20.0
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        gating = base.to(dt)
        w, ids = ops.grouped_topk_cpu(gating.cuda(), gating.cuda(), topk, True, 1, 1, 0, None, None)
        ow, oids = routing.grouped_topk(gating, topk, True, 1, 1)
        assert torch.equal(ids.cpu(), oids), dt
        assert torch.allclose(w.cpu(), ow, rtol=2e-5, atol=1e-7)


def test_grouped_topk_ties_and_odd_shapes_bit_exact(ops):
    """The picks run on 64-bit (value, ~index) keys (topk.hip): tie-heavy logits (a handful of levels, +0 and -0 among them, or all
    equal), expert counts that are not powers of two, every group / top-k split, selections that run out of selected-group experts
    -- ids must equal the oracle's (larger value, then lower index) bit for bit.  tools/fuzz_topk.py is the long form."""
    import random
    rng = random.Random(99)
    for it in range(60):
        E = rng.choice([4, 8, 24, 64, 96, 128, 160, 256, 384, 1024])
        G = rng.choice([g for g in (1, 2, 3, 4, 8, 16, 32, 64) if E % g == 0 and g <= E])
        topk_group, topk = rng.randint(1, G), rng.randint(1, min(E, 12))
        M = rng.choice([1, 5, 17, 64])
        dt = rng.choice([torch.float32, torch.bfloat16, torch.float16])
        g = torch.Generator().manual_seed(rng.randrange(1 << 30))
        if rng.random() < 0.2:
            gating = torch.zeros(M, E)
        else:
            levels = torch.tensor([-2.5, -1.0, -0.0, 0.0, 0.5, 0.5, 3.0])
            gating = levels[torch.randint(0, len(levels), (M, E), generator=g)]
        gating = gating.to(dt)
        ow, oids = routing.grouped_topk(gating, topk, True, G, topk_group)
        w, ids = ops.grouped_topk_cpu(gating.cuda(), gating.cuda(), topk, True, G, topk_group, 0, None, None)
        assert torch.equal(ids.cpu().to(torch.int32), oids), f"case {it}: E={E} G={G} topk_group={topk_group} topk={topk} {dt}"
        assert torch.allclose(torch.nan_to_num(w.cpu()), torch.nan_to_num(ow), rtol=2e-5, atol=1e-6)
