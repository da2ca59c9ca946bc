"""Legacy import style of the reference harness: `from sgl_kernel.common_ops import <op>`
(/root/reference/test_gemm.py:2-3, /root/reference/test_moe_fp8.py:4-5).  Thin aliases of torch.ops.sgl_kernel."""
import torch

from . import _ops  # noqa: F401

convert_weight_packed = torch.ops.sgl_kernel.convert_weight_packed
fused_experts_cpu = torch.ops.sgl_kernel.fused_experts_cpu
shared_expert_cpu = torch.ops.sgl_kernel.shared_expert_cpu
weight_packed_linear = torch.ops.sgl_kernel.weight_packed_linear
fp8_scaled_mm_cpu = torch.ops.sgl_kernel.fp8_scaled_mm_cpu
per_token_quant_int8_cpu = torch.ops.sgl_kernel.per_token_quant_int8_cpu
int8_scaled_mm_cpu = torch.ops.sgl_kernel.int8_scaled_mm_cpu
int8_scaled_mm_with_quant = torch.ops.sgl_kernel.int8_scaled_mm_with_quant
silu_and_mul_cpu = torch.ops.sgl_kernel.silu_and_mul_cpu
rmsnorm_cpu = torch.ops.sgl_kernel.rmsnorm_cpu
fused_add_rmsnorm_cpu = torch.ops.sgl_kernel.fused_add_rmsnorm_cpu
grouped_topk_cpu = torch.ops.sgl_kernel.grouped_topk_cpu
biased_grouped_topk_cpu = torch.ops.sgl_kernel.biased_grouped_topk_cpu
qkv_proj_with_rope = torch.ops.sgl_kernel.qkv_proj_with_rope
flash_attn_varlen_func = torch.ops.sgl_kernel.flash_attn_varlen_func
bmm_cpu = torch.ops.sgl_kernel.bmm_cpu
convert_scale_packed = torch.ops.sgl_kernel.convert_scale_packed
mxfp4_scaled_mm_cpu = torch.ops.sgl_kernel.mxfp4_scaled_mm_cpu

# tensor-parallel collectives (/root/reference/test_allreduce.py:82-87,103-105): RCCL / gloo through torch.distributed
from .collectives import initialize, shm_allgather, shm_allreduce  # noqa: E402,F401
