"""CPU restatement of the reference's oracle for qkv_proj_with_rope (MLA "absorbed" q/k/v projection + RMSNorm + RoPE).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/ to check the HIP path, never by the product.
Follows /root/reference/test_absorb.py: layernorm :20-25, _rotate_gptj :27-31, per_token_quant_int8 :33-40,
native_w8a8_per_token_matmul :42-47, rotary_emb :49-63, native_torch :65-87, native_torch_int8 :89-109.  Rounding points
are the reference's: every matmul / bmm returns bf16, the norms and the rotation compute in fp32 and round once.
Pinned by tests/golden/absorb_*.safetensors (generated from the reference's own functions, tests/golden/make_golden.py).
"""
import torch

from .gemm import quant_int8_rowwise


def rmsnorm(x, weight, eps=1e-6):
    """test_absorb.py:20-25 (despite its name `layernorm` it is an RMS norm)."""
    xf = x.float()
    xf = xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)
    return (xf * weight).to(x.dtype)


def rotate_gptj(x):
    """test_absorb.py:27-31: (x0, x1, x2, x3, ...) -> (-x1, x0, -x3, x2, ...)."""
    return torch.stack((-x[..., 1::2], x[..., ::2]), dim=-1).flatten(-2)


def rope(q_pe, k_pe, pos, cos_sin_cache):
    """test_absorb.py:49-63: cache row = [cos(0..d/2) | sin(0..d/2)], each entry applied to an adjacent pair."""
    dt = q_pe.dtype
    cs = cos_sin_cache.float()[pos]
    cos, sin = cs.chunk(2, dim=-1)
    cos = cos.repeat_interleave(2, dim=-1).unsqueeze(-2)
    sin = sin.repeat_interleave(2, dim=-1).unsqueeze(-2)
    q, k = q_pe.float(), k_pe.float()
    return (q * cos + rotate_gptj(q) * sin).to(dt), (k * cos + rotate_gptj(k) * sin).to(dt)


def _w8a8(a, wq, ws):
    """test_absorb.py:33-47: per-token quantisation (floor 1e-7), As * (Aq . Bq^T) * Bs in fp32, bf16 result."""
    aq, a_s = quant_int8_rowwise(a, floor=1e-7)
    return (a_s.view(-1, 1) * (aq.float() @ wq.float().t()) * ws.float().view(1, -1)).to(torch.bfloat16)


def qkv_proj_with_rope(hidden, q_a, q_b, kv_a, w_kc, norm1, norm2, pos, cos_sin_cache, eps=1e-6, scales=None):
    """w_kc is [H, kv_lora_rank, qk_nope] (the layout the operator receives, test_absorb.py:145); scales = (s_qa, s_qb,
    s_kva) per output row for the int8 variant (weights then int8).  Returns (q_input [B,H,R+rope], k_input [B,1,R+rope],
    v_input [B,1,R])."""
    H, R, nope = w_kc.shape
    B = hidden.shape[0]
    if scales is None:
        lin = lambda x, w, s: torch.matmul(x, w.t())
        s_qa = s_qb = s_kva = None
    else:
        lin = _w8a8
        s_qa, s_qb, s_kva = scales
    q = rmsnorm(lin(hidden, q_a, s_qa), norm1, eps)
    q = lin(q, q_b, s_qb).view(B, H, -1)
    q_nope, q_pe = q[..., :nope], q[..., nope:]
    q_nope_out = torch.bmm(q_nope.transpose(0, 1), w_kc.transpose(1, 2)).transpose(0, 1)      # [B, H, R]
    latent = lin(hidden, kv_a, s_kva)
    v_input = rmsnorm(latent[..., :R].contiguous(), norm2, eps).unsqueeze(1)
    k_pe = latent[..., R:].unsqueeze(1)
    q_pe, k_pe = rope(q_pe, k_pe, pos, cos_sin_cache)
    q_input = torch.cat([q_nope_out, q_pe], dim=-1)
    k_input = torch.cat([v_input, k_pe], dim=-1)
    return q_input, k_input, v_input
