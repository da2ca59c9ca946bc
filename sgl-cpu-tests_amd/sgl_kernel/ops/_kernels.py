"""Oldest import style of the reference harness: `from sgl_kernel.ops._kernels import silu_and_mul_cpu`
(/root/reference/test_activation.py:10) — out-parameter form `silu_and_mul_cpu(out, x)`."""
import torch

from .. import _ops  # noqa: F401

silu_and_mul_cpu = torch.ops.sgl_kernel.silu_and_mul_cpu
