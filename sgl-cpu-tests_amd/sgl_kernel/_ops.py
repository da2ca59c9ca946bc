"""torch.library registration of the sgl_kernel operators (reference signatures) over the sglk C-ABI."""
import ctypes

import torch

from . import _lib

_DEF = torch.library.Library("sgl_kernel", "DEF")

_WTYPE = {torch.bfloat16: _lib.W_BF16, torch.float8_e4m3fn: _lib.W_FP8_E4M3, torch.int8: _lib.W_INT8}


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError(
            "sgl_kernel: no GPU visible. These operators run hand-written gfx950 HIP kernels only; "
            "there is no CPU fallback (host tensors are staged through the GPU).")
    return torch.device("cuda", torch.cuda.current_device())


def _staged(fn):
    """CPU-key implementation: host buffers in -> device -> HIP kernel -> host buffers out.

    Keeps the reference scripts (which build CPU tensors) working unchanged.  In-place semantics are
    preserved by copying mutated arguments back.  This is a PCIe round trip, not a compute fallback.
    """
    def wrapper(*args):
        dev = _require_gpu()
        moved = [a.to(dev) if isinstance(a, torch.Tensor) else a for a in args]
        out = fn(*moved)
        for a, m in zip(args, moved):
            if isinstance(a, torch.Tensor) and fn._mutates(a, args):
                a.copy_(m)
        if isinstance(out, torch.Tensor):
            for a, m in zip(args, moved):
                if isinstance(a, torch.Tensor) and out is m:
                    return a
            return out.cpu()
        if isinstance(out, tuple):
            return tuple(o.cpu() if isinstance(o, torch.Tensor) else o for o in out)
        return out
    return wrapper


def _impl(name, fn, mutates=lambda a, args: False):
    fn._mutates = mutates
    _DEF.impl(name, fn, "CUDA")
    _DEF.impl(name, _staged(fn), "CPU")


# ------------------------------------------------------------------------------------------------------
# convert_weight_packed            /root/reference/bench_moe.py:26-27,43-44; test_gemm.py:24
# ------------------------------------------------------------------------------------------------------
_DEF.define("convert_weight_packed(Tensor weight) -> Tensor")


def _pack_supported(rows, cols, dtype):
    tc = 32 if dtype == torch.bfloat16 else 64
    return rows % 16 == 0 and cols % tc == 0


def convert_weight_packed(weight):
    if weight.dtype not in _WTYPE:
        raise RuntimeError(f"convert_weight_packed: unsupported dtype {weight.dtype}")
    if weight.dim() not in (2, 3):
        raise RuntimeError("convert_weight_packed: expect a 2-D [N,K] or 3-D [E,N,K] weight")
    w = weight.contiguous()
    rows, cols = w.shape[-2], w.shape[-1]
    batch = w.shape[0] if w.dim() == 3 else 1
    out = torch.empty_like(w)
    if not _pack_supported(rows, cols, w.dtype):
        # shapes the MFMA tile order cannot hold stay row-major; the kernels pick the generic path for exactly
        # these shapes, so "packed" stays a pure function of (shape, dtype)
        out.copy_(w)
        return out
    rc = _lib.lib().sglk_pack_weight(_ptr(w), _ptr(out), batch, rows, cols, _WTYPE[w.dtype], _stream(w))
    _lib.check(rc, "convert_weight_packed")
    return out


_impl("convert_weight_packed", convert_weight_packed)


# ------------------------------------------------------------------------------------------------------
# fused_experts_cpu                14-arg: /root/reference/bench_moe.py:113-130, test_moe_fp8_ext.py:118
#                                  13-arg: /root/reference/test_moe.py:79-92 (CPUQuantMethod)
# ------------------------------------------------------------------------------------------------------
_DEF.define(
    "fused_experts_cpu(Tensor(a!) hidden_states, Tensor w1, Tensor w2, Tensor topk_weights, Tensor topk_ids, "
    "bool inplace, bool use_int8_w8a8, bool use_fp8_w8a16, Tensor? w1_scale, Tensor? w2_scale, "
    "int[]? block_size, Tensor? a1_scale, Tensor? a2_scale, bool is_vnni) -> Tensor")
_DEF.define(
    "fused_experts_cpu.method(Tensor(a!) hidden_states, Tensor w1, Tensor w2, Tensor topk_weights, Tensor topk_ids, "
    "bool inplace, int moe_comp_method, Tensor? w1_scale, Tensor? w2_scale, "
    "int[]? block_size, Tensor? a1_scale, Tensor? a2_scale, bool is_vnni) -> Tensor")

# measurement hook (bench.py): when set to a sglk_stage_timer handle, every fused_experts call records stage events
_stage_timer = None


def set_stage_timer(handle):
    global _stage_timer
    _stage_timer = handle


# sglang.srt.layers.amx_utils.CPUQuantMethod values (shim in sgl-cpu-tests_amd/sglang)
UNQUANT, INT8_W8A8, FP8_W8A16 = 0, 1, 2


def _fused_experts(hidden_states, w1, w2, topk_weights, topk_ids, inplace, method, w1_scale, w2_scale,
                   block_size, a1_scale, a2_scale, is_vnni):
    if hidden_states.dim() != 2 or w1.dim() != 3 or w2.dim() != 3:
        raise RuntimeError("fused_experts: expect hidden [M,K], w1 [E,2N,K], w2 [E,K,N]")
    if hidden_states.dtype != torch.bfloat16:
        raise RuntimeError(f"fused_experts: hidden_states must be bfloat16 (got {hidden_states.dtype})")
    M, K = hidden_states.shape
    E, N2, K1 = w1.shape
    N = N2 // 2
    if K1 != K or tuple(w2.shape) != (E, K, N) or N2 != 2 * N:
        raise RuntimeError(f"fused_experts: shape mismatch hidden {tuple(hidden_states.shape)} "
                           f"w1 {tuple(w1.shape)} w2 {tuple(w2.shape)}")
    if topk_weights.shape != topk_ids.shape or topk_ids.dim() != 2 or topk_ids.shape[0] != M:
        raise RuntimeError("fused_experts: topk_weights/topk_ids must both be [M, topk]")
    if a1_scale is not None or a2_scale is not None:
        raise RuntimeError("fused_experts: static activation scales (a1_scale/a2_scale) are not supported")
    topk = topk_ids.shape[1]
    wdtype = {UNQUANT: torch.bfloat16, INT8_W8A8: torch.int8, FP8_W8A16: torch.float8_e4m3fn}.get(int(method))
    if wdtype is None:
        raise RuntimeError(f"fused_experts: unknown quant method {method}")
    if w1.dtype != wdtype or w2.dtype != wdtype:
        raise RuntimeError(f"fused_experts: weights must be {wdtype} for this mode (got {w1.dtype}, {w2.dtype})")
    if hidden_states.stride(1) != 1:
        hidden_states_c = hidden_states.contiguous()
    else:
        hidden_states_c = hidden_states
    topk_weights = topk_weights.to(torch.float32).contiguous()
    topk_ids = topk_ids.to(torch.int32).contiguous()
    w1, w2 = w1.contiguous(), w2.contiguous()
    bn, bk = 0, 0
    if int(method) == FP8_W8A16:
        if block_size is None or len(block_size) != 2:
            raise RuntimeError("fused_experts: fp8 needs block_size = [block_n, block_k]")
        bn, bk = int(block_size[0]), int(block_size[1])
    if int(method) != UNQUANT:
        if w1_scale is None or w2_scale is None:
            raise RuntimeError("fused_experts: quantised modes need w1_scale and w2_scale")
        w1_scale = w1_scale.to(torch.float32).contiguous()
        w2_scale = w2_scale.to(torch.float32).contiguous()

    out = hidden_states_c if (inplace and hidden_states_c is hidden_states) else torch.empty_like(hidden_states_c)
    L = _lib.lib()
    wtype = _WTYPE[wdtype]
    ws_bytes = L.sglk_fused_experts_workspace_bytes(M, N, K, E, topk, wtype)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=hidden_states.device)
    args = _lib.FusedExpertsArgs(
        hidden=hidden_states_c.data_ptr(), hidden_stride=hidden_states_c.stride(0),
        out=out.data_ptr(), out_stride=out.stride(0),
        w1=w1.data_ptr(), w2=w2.data_ptr(),
        w1_scale=w1_scale.data_ptr() if w1_scale is not None else None,
        w2_scale=w2_scale.data_ptr() if w2_scale is not None else None,
        topk_weights=topk_weights.data_ptr(), topk_ids=topk_ids.data_ptr(),
        M=M, N=N, K=K, E=E, topk=topk, wtype=wtype,
        packed=1 if (is_vnni and _pack_supported(2 * N, K, wdtype) and _pack_supported(K, N, wdtype)) else 0,
        block_n=bn, block_k=bk, workspace=ws.data_ptr(), workspace_bytes=ws_bytes, stage_timer=_stage_timer)
    rc = L.sglk_fused_experts(ctypes.byref(args), _stream(hidden_states))
    _lib.check(rc, "fused_experts_cpu")
    if inplace and out is not hidden_states:
        hidden_states.copy_(out)
        return hidden_states
    return out


def fused_experts_cpu(hidden_states, w1, w2, topk_weights, topk_ids, inplace, use_int8_w8a8, use_fp8_w8a16,
                      w1_scale, w2_scale, block_size, a1_scale, a2_scale, is_vnni):
    if use_int8_w8a8 and use_fp8_w8a16:
        raise RuntimeError("fused_experts: use_int8_w8a8 and use_fp8_w8a16 are mutually exclusive")
    method = INT8_W8A8 if use_int8_w8a8 else (FP8_W8A16 if use_fp8_w8a16 else UNQUANT)
    return _fused_experts(hidden_states, w1, w2, topk_weights, topk_ids, inplace, method, w1_scale, w2_scale,
                          block_size, a1_scale, a2_scale, is_vnni)


def fused_experts_cpu_method(hidden_states, w1, w2, topk_weights, topk_ids, inplace, moe_comp_method,
                             w1_scale, w2_scale, block_size, a1_scale, a2_scale, is_vnni):
    return _fused_experts(hidden_states, w1, w2, topk_weights, topk_ids, inplace, int(moe_comp_method),
                          w1_scale, w2_scale, block_size, a1_scale, a2_scale, is_vnni)


def _mut_inplace(a, args):
    # hidden_states (arg 0) is overwritten when inplace (arg 5) is true
    return a is args[0] and bool(args[5])


_impl("fused_experts_cpu", fused_experts_cpu, _mut_inplace)
_impl("fused_experts_cpu.method", fused_experts_cpu_method, _mut_inplace)
