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


E2M1_VALUES = (0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0)
E2M1_BOUNDS = (0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0)


def mxfp4_quantize(x, block=32):
    """MXFP4QuantizeUtil.quantize (/root/reference/test_mxfp4.py:23-64): per 32-wide block, scale exponent
    ceil(log2(amax / 6)) clamped at -127, values to the nearest E2M1 code by the bounds table (ties go down), element 2i
    in the low nibble of byte i, scale stored as exponent + 127."""
    shape = x.shape
    xb = x.reshape(-1, block)
    amax = xb.abs().max(dim=-1, keepdim=True).values
    e = torch.ceil(torch.maximum(torch.log2(amax / 6.0), torch.tensor(-127.0)))
    q = (xb / torch.exp2(e)).reshape(shape)
    sign_bit = (2 - torch.sign(q)) // 2
    code = (q.abs().unsqueeze(-1) > torch.tensor(E2M1_BOUNDS)).sum(dim=-1)
    nib = (sign_bit * 8 + code).to(torch.uint8)
    packed = (nib[..., 1::2] << 4) + nib[..., 0::2]
    return packed, (e + 127).to(torch.uint8)


def mxfp4_dequant(wq, scales, block=32):
    """MXFP4QuantizeUtil.dequantize (/root/reference/test_mxfp4.py:66-127) in fp32: nibble -> +-E2M1 value, times
    2^(scale - 127) per 32 consecutive elements of a row."""
    lo, hi = wq & 0x0F, (wq >> 4) & 0x0F
    nib = torch.stack([lo, hi], dim=-1).reshape(*wq.shape[:-1], wq.shape[-1] * 2)
    sign = 1.0 - 2.0 * ((nib & 8) >> 3).float()
    val = sign * torch.tensor(E2M1_VALUES)[(nib & 7).long()]
    sc = torch.exp2(scales.float() - 127.0).reshape(-1, 1)
    return (val.reshape(-1, block) * sc).reshape(nib.shape)


def mxfp4_scaled_mm(x, wq, scales, bias=None):
    """mxfp4_scaled_mm_cpu's expectation (/root/reference/test_mxfp4.py:166-168): fp32 matmul against the dequantised
    weights (exactly representable in bf16), + bias; fp32 here, the caller rounds."""
    out = x.float() @ mxfp4_dequant(wq, scales).t()
    return out if bias is None else out + bias.float().view(1, -1)


def scale_packed_order(scales):
    """convert_scale_packed's layout as the reference checks it (/root/reference/test_mxfp4.py:186):
    [N, K/32] -> [N/32][K/32][32]."""
    n, kb = scales.shape
    return scales.view(n // 32, 32, kb).transpose(1, 2).contiguous()
