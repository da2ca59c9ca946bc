"""ctypes binding of oracle/_build/liboracle.so (plain-C restatement) — test infrastructure only."""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    """Compile oracle/c/*.c with gcc (Makefile in this directory)."""
    if force or not os.path.exists(_LIB):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []))
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB)
        _lib.sglk_oracle_fused_experts_fp8.restype = ctypes.c_int
        _lib.sglk_oracle_num_threads.restype = ctypes.c_int
    return _lib


def num_threads():
    return lib().sglk_oracle_num_threads()


def set_threads(n):
    """OpenMP threads of the following calls (bench.py's cpu_baseline: the cores of one NUMA node)."""
    lib().sglk_oracle_set_threads(int(n))


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def fused_experts_fp8(a, w1_fp8, w2_fp8, w1_scale, w2_scale, block, topk_weight, topk_ids):
    """Same contract as oracle.moe.fused_experts_fp8; CPU tensors in, fp32 [M,K] out."""
    assert a.dtype == torch.bfloat16 and w1_fp8.dtype == torch.float8_e4m3fn
    a, w1_fp8, w2_fp8 = a.contiguous(), w1_fp8.contiguous(), w2_fp8.contiguous()
    w1_scale, w2_scale = w1_scale.float().contiguous(), w2_scale.float().contiguous()
    topk_weight = topk_weight.float().contiguous()
    topk_ids = topk_ids.to(torch.int32).contiguous()
    M, K = a.shape
    E, N2, _ = w1_fp8.shape
    N = N2 // 2
    topk = topk_ids.shape[1]
    out = torch.empty(M, K, dtype=torch.float32)
    rc = lib().sglk_oracle_fused_experts_fp8(
        _p(a), M, N, K, E, topk, _p(w1_fp8), _p(w2_fp8), _p(w1_scale), _p(w2_scale),
        int(block[0]), int(block[1]), _p(topk_weight), _p(topk_ids), _p(out))
    if rc != 0:
        raise MemoryError("sglk_oracle_fused_experts_fp8 failed (allocation)")
    return out
