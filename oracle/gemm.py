"""Torch restatement of the reference's dense-GEMM oracles (test infrastructure), fp32 math on CPU tensors."""
import torch

from .moe import dequant_block_fp8, quant_int8_rowwise


def linear_bf16(x, w, bias=None):
    """x [M,K] . w [N,K]^T (+ bias rounded to bf16 first, as /root/reference/test_gemm.py:15-20 does) -> fp32."""
    out = x.float() @ w.float().t()
    if bias is not None:
        out = out + bias.bfloat16().float()
    return out


def fp8_scaled_mm(x, w_fp8, scales, block, bias=None):
    """x . (w_fp8 * blockscale)^T + bias in fp32; /root/reference/test_gemm_fp8.py:22-49 (its own oracle rounds the
    dequantised weight and the product to bf16; this is the fp32 formulation of the same product)."""
    w = dequant_block_fp8(w_fp8, scales, block[0], block[1])
    out = x.float() @ w.t()
    if bias is not None:
        out = out + bias.float()
    return out


def per_token_quant_int8(x):
    """/root/reference/test_gemm_int8.py:14-22: floor 1e-10.  Returns (int8 [M,K], f32 scale [M])."""
    q, s = quant_int8_rowwise(x, floor=1e-10)
    return q, s.reshape(-1)


def int8_scaled_mm(xq, x_scale, wq, w_scale, bias=None):
    """As * (Aq . Bq^T) * Bs + bias, float accumulation; /root/reference/test_gemm_int8.py:25-47."""
    out = (xq.float() @ wq.float().t()) * x_scale.float().view(-1, 1) * w_scale.float().view(1, -1)
    if bias is not None:
        out = out + bias.float().view(1, -1)
    return out


def bmm(mat1, mat2):
    """bmm_cpu (/root/reference/test_bmm_fp8.py:57,67): out[b] = mat1[b] @ mat2[b]^T with mat2 [B, N, K]; fp32 here, the
    caller rounds to bf16."""
    return torch.bmm(mat1.float(), mat2.float().transpose(1, 2))
