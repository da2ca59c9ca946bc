"""Torch restatement of the reference's activation / norm oracles (test infrastructure)."""
import torch
import torch.nn.functional as F


def silu_and_mul(x):
    """F.silu(x[..., :d]) * x[..., d:] in the input dtype; /root/reference/test_activation.py:14-16."""
    d = x.shape[-1] // 2
    return F.silu(x[..., :d]) * x[..., d:]


def rmsnorm(x, weight, eps=1e-6, residual=None):
    """/root/reference/test_norm.py:15-33: fp32 variance, normalised value cast to the I/O dtype, times weight."""
    dt = x.dtype
    x = x.float()
    if residual is not None:
        x = x + residual.float()
        residual = x.to(dt)
    x = x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + eps)
    out = x.to(dt) * weight
    return out if residual is None else (out, residual)
