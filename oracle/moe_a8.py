"""Oracle of the opt-in "a8" mode (sgl-cpu-tests_amd/csrc/moe_gemm_a8.hip) -- TEST INFRASTRUCTURE ONLY.

The a8 mode is NOT the reference's arithmetic (the reference op is W8A16: /root/reference/bench_moe.py:113-130 keeps the
activations in bf16; its oracle is /root/reference/test_moe_fp8_ext.py:70-91, restated in oracle/moe.py).  This file
restates what the a8 KERNELS compute, so that they can be held to a tight tolerance of their own:

  * activations (hidden, and SiLU*mul's output ic1) are quantised per row x 128-wide block to e4m3 (round to nearest
    even) with a power-of-two scale 2^(sb-127): sb = the smallest exponent with amax / 2^e <= 448, clamped to [1, 253]
    (e8m0_for_amax below mirrors the kernel's integer rule bit for bit);
  * everything else as in the reference's oracle: weights dequantised with their fp32 block scales, exact sums (float64
    here), ic2 = bf16(topk_weight * y) per slot, fp32 sum over the slots in slot order, one bf16 rounding.
The kernels differ from this only by fp32 accumulation order (~1e-6 relative) and by the rare rounding-boundary flips
that follow from it.
"""
import numpy as np
import torch


def e8m0_for_amax(amax):
    """amax: float32 tensor >= 0 -> int32 tensor of E8M0 bytes (same integer arithmetic as the kernel)."""
    u = amax.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    sb = (u >> 23) - 8 + ((u & 0x7FFFFF) > 0x600000).to(torch.int64)
    return sb.clamp(1, 253).to(torch.int32)


def quant_block128(x):
    """x [R, C] float32 (C % 128 == 0) -> (dequantised values float32 [R, C], e4m3 bytes uint8 [R, C], sb int32 [R, C/128])."""
    R, C = x.shape
    assert C % 128 == 0
    xb = x.float().reshape(R, C // 128, 128)
    amax = xb.abs().amax(dim=-1)
    sb = e8m0_for_amax(amax)
    inv = torch.ldexp(torch.ones_like(amax), 127 - sb)            # 2^(127 - sb), exact
    q8 = (xb * inv.unsqueeze(-1)).to(torch.float8_e4m3fn)          # round to nearest even; |.| <= 448 by construction
    deq = q8.float() * torch.ldexp(torch.ones_like(amax), sb - 127).unsqueeze(-1)
    return deq.reshape(R, C), q8.view(torch.uint8).reshape(R, C), sb


def packed_k_order(C):
    """Storage order of a quantised row: position p of every 64 group holds k = perm[p] (moe_gemm_a8.hip header)."""
    perm = np.empty(64, dtype=np.int64)
    for h in range(2):
        for q in range(32):
            k = 16 * h + (q if q < 8 else 32 + q - 8 if q < 16 else 8 + q - 16 if q < 24 else 40 + q - 24)
            perm[32 * h + q] = k
    assert sorted(perm.tolist()) == list(range(64))
    return (np.arange(0, C, 64)[:, None] + perm[None, :]).reshape(-1)


def _dequant_weight(w_fp8, scale, bn, bk):
    R, C = w_fp8.shape
    s = scale.double().repeat_interleave(bn, 0)[:R].repeat_interleave(bk, 1)[:, :C]
    return w_fp8.float().double() * s


def fused_experts_a8(a, w1_fp8, w2_fp8, w1_scale, w2_scale, block, topk_weight, topk_ids):
    """Same contract as oracle.moe.fused_experts_fp8 (CPU tensors in, float32 [M, K] out), a8 arithmetic."""
    bn, bk = int(block[0]), int(block[1])
    assert bk == 128
    M, K = a.shape
    E, N2, _ = w1_fp8.shape
    N = N2 // 2
    topk = topk_ids.shape[1]
    x_deq, _, _ = quant_block128(a.float())
    ids = topk_ids.to(torch.int64)
    tw = topk_weight.float()
    ic2 = torch.zeros(M, topk, K, dtype=torch.float32)
    for e in torch.unique(ids[(ids >= 0) & (ids < E)]).tolist():
        tok, slot = torch.nonzero(ids == e, as_tuple=True)
        W1 = _dequant_weight(w1_fp8[e], w1_scale[e], bn, bk)
        W2 = _dequant_weight(w2_fp8[e], w2_scale[e], bn, bk)
        gu = x_deq[tok].double() @ W1.t()
        g, u = gu[:, :N], gu[:, N:]
        h = (g / (1.0 + torch.exp(-g)) * u).float()
        h_deq, _, _ = quant_block128(h)
        y = (h_deq.double() @ W2.t()).float()
        ic2[tok, slot] = (tw[tok, slot].unsqueeze(-1) * y).bfloat16().float()
    return ic2.sum(dim=1).bfloat16().float()
