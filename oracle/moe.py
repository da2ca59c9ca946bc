"""Torch restatement of the reference's fused-MoE / shared-expert oracles (test infrastructure).

All math in fp32 on CPU tensors, exactly as the reference oracles do.  Written per expert over
index lists instead of boolean masks; results are identical because every (token, slot) row is an
independent dot-product chain.
"""
import torch


def silu_mul(x):
    """SiLU(x[..., :d]) * x[..., d:]   -- /root/reference/test_moe_fp8_ext.py:18-20."""
    d = x.shape[-1] // 2
    gate, up = x[..., :d], x[..., d:]
    return gate * torch.sigmoid(gate) * up


def dequant_block_fp8(w, scale, block_n, block_k):
    """fp8 weight [E,R,C] (or [R,C]) x per-block f32 scale [E,R/bn,C/bk] -> f32.

    /root/reference/test_moe_fp8_ext.py:22-25 (scaled_weight).  The scale of element (r, c) is
    scale[r // bn, c // bk].
    """
    squeeze = w.dim() == 2
    if squeeze:
        w, scale = w[None], scale[None]
    E, R, C = w.shape
    s = scale.float().repeat_interleave(block_n, dim=1)[:, :R].repeat_interleave(block_k, dim=2)[:, :, :C]
    out = w.float() * s
    return out[0] if squeeze else out


def fused_experts_f32(a, w1, w2, topk_weight, topk_ids):
    """Routed-expert MLP with fp32 weights.  a [M,K] any float dtype, w1 [E,2N,K] f32, w2 [E,K,N] f32.

    out[m] = sum_j topk_weight[m,j] * ( silu_mul(a[m] @ w1[e].T) @ w2[e].T ),  e = topk_ids[m,j];
    slots whose id is outside [0,E) (the -1 padding) contribute nothing.
    /root/reference/test_moe_fp8_ext.py:70-91, /root/reference/test_moe_offloading_cpu.py:33-52.
    Returns fp32 [M,K].
    """
    M, K = a.shape
    E = w1.shape[0]
    topk = topk_ids.shape[1]
    x = a.float()
    slot_out = torch.zeros(M * topk, K, dtype=torch.float32)
    flat_ids = topk_ids.reshape(-1).long()
    for e in range(E):
        slots = (flat_ids == e).nonzero(as_tuple=True)[0]
        if slots.numel() == 0:
            continue
        rows = x[slots // topk]
        h = silu_mul(rows @ w1[e].float().t())
        slot_out[slots] = h @ w2[e].float().t()
    weighted = slot_out.view(M, topk, K) * topk_weight.float().view(M, topk, 1)
    return weighted.sum(dim=1)


def fused_experts_fp8(a, w1_fp8, w2_fp8, w1_scale, w2_scale, block, topk_weight, topk_ids):
    """fp8-weight (W8A16) routed experts: dequantise with block scales, then fused_experts_f32."""
    bn, bk = block
    w1 = dequant_block_fp8(w1_fp8, w1_scale, bn, bk)
    w2 = dequant_block_fp8(w2_fp8, w2_scale, bn, bk)
    return fused_experts_f32(a, w1, w2, topk_weight, topk_ids)


def quant_int8_rowwise(x, floor=1e-7):
    """Per-row symmetric int8: scale = max(|row|, floor)/127, q = round(x / scale).

    /root/reference/test_moe_int8.py:23-31 (floor 1e-7); /root/reference/test_gemm_int8.py:14-22 uses
    floor 1e-10.  Returns (int8 [.., C], f32 scale [.., 1]).
    """
    x = x.float()
    amax = x.abs().amax(dim=-1, keepdim=True).clamp_min(floor)
    q = torch.round(x * (127.0 / amax)).to(torch.int8)
    return q, amax / 127.0


def fused_experts_int8(a, w1_q, w2_q, w1_s, w2_s, topk_weight, topk_ids):
    """w8a8 routed experts, dynamic per-token activation quant before each GEMM.

    /root/reference/test_moe_int8.py:59-94.  w1_q [E,2N,K] int8, w1_s [E,2N] f32 (per out channel).
    """
    M, K = a.shape
    E = w1_q.shape[0]
    topk = topk_ids.shape[1]
    aq, a_s = quant_int8_rowwise(a)
    slot_out = torch.zeros(M * topk, K, dtype=torch.float32)
    flat_ids = topk_ids.reshape(-1).long()
    for e in range(E):
        slots = (flat_ids == e).nonzero(as_tuple=True)[0]
        if slots.numel() == 0:
            continue
        tok = slots // topk
        g1 = (aq[tok].float() @ w1_q[e].float().t()) * a_s[tok] * w1_s[e].float().view(1, -1)
        h = silu_mul(g1)
        hq, h_s = quant_int8_rowwise(h)
        slot_out[slots] = (hq.float() @ w2_q[e].float().t()) * h_s * w2_s[e].float().view(1, -1)
    weighted = slot_out.view(M, topk, K) * topk_weight.float().view(M, topk, 1)
    return weighted.sum(dim=1)


def softmax_topk(score, topk, renormalize):
    """/root/reference/test_moe.py:26-30."""
    p = torch.softmax(score, dim=-1, dtype=torch.float32)
    w, ids = torch.topk(p, topk)
    if renormalize:
        w = w / w.sum(dim=-1, keepdim=True)
    return w, ids.to(torch.int32)


def shared_expert_f32(a, w1, w2, fused_out, routed_scaling_factor):
    """Dense SiLU-MLP + fused_out * rsf in fp32; /root/reference/test_moe_fp8_ext.py:52-56,
    /root/reference/test_shared_experts.py:34-40."""
    h = silu_mul(a.float() @ w1.float().t())
    return h @ w2.float().t() + fused_out.float() * routed_scaling_factor


def shared_expert_int8(a, w1_q, w2_q, w1_s, w2_s, fused_out, routed_scaling_factor):
    """/root/reference/test_shared_experts.py:42-53."""
    aq, a_s = quant_int8_rowwise(a)
    g1 = (aq.float() @ w1_q.float().t()) * a_s * w1_s.float().view(1, -1)
    hq, h_s = quant_int8_rowwise(silu_mul(g1))
    out = (hq.float() @ w2_q.float().t()) * h_s * w2_s.float().view(1, -1)
    return out + fused_out.float() * routed_scaling_factor


def allclose_ref(a, b, dtype=None):
    """The reference's pass predicate: allclose(a, b, rtol=atol=pres[a.dtype]); /root/reference/utils.py:3-13."""
    pres = {torch.bfloat16: 1e-2, torch.float16: 1e-3, torch.float32: 1e-5}
    tol = pres[dtype or a.dtype]
    return torch.allclose(a.float(), b.float(), rtol=tol, atol=tol)
