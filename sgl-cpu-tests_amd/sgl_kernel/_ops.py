"""torch.library registration of the sgl_kernel operators (reference signatures) over the sglk C-ABI."""
import ctypes
import os

import torch

from . import _lib

_DEF = torch.library.Library("sgl_kernel", "DEF")

_WTYPE = {torch.bfloat16: _lib.W_BF16, torch.float8_e4m3fn: _lib.W_FP8_E4M3, torch.int8: _lib.W_INT8}


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


_ws_cache = {}


def _workspace(nbytes, device):
    """Scratch for one call.  Reused per (device, stream): work on one stream is ordered, so the next call on the same
    stream may overwrite it; other streams get their own buffer.  Grows monotonically.

    Under hipGraph capture every call gets a FRESH buffer that is not cached: its address is baked into the graph, its
    lifetime is the graph's private pool, and two graphs captured on one stream (which may be replayed concurrently on
    different streams) must not share scratch -- nor may a later eager call's larger request drop a buffer a graph uses."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        if len(_ws_cache) > 64:
            _ws_cache.clear()
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


class _Aux:
    """A second HIP stream + two events owned by this layer (sglk_aux_create), handed to sglk_fused_experts so that the
    tail tiles of a mid-size batch run beside the big launches.  One per (device, caller stream)."""
    __slots__ = ("stream", "ev0", "ev1")

    def __init__(self):
        st, e0, e1 = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(_lib.lib().sglk_aux_create(ctypes.byref(st), ctypes.byref(e0), ctypes.byref(e1)), "aux_create")
        self.stream, self.ev0, self.ev1 = st.value, e0.value, e1.value

    def __del__(self):
        try:
            _lib.lib().sglk_aux_destroy(self.stream, self.ev0, self.ev1)
        except Exception:
            pass


_aux_cache = {}


def _aux(device):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    a = _aux_cache.get(key)
    if a is None:
        if len(_aux_cache) > 64:
            _aux_cache.clear()
        a = _aux_cache[key] = _Aux()
    return a


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()


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
            def back(o):
                if not isinstance(o, torch.Tensor):
                    return o
                for a, m in zip(args, moved):
                    if isinstance(a, torch.Tensor) and o is m:
                        return a          # an in-place result: hand back the caller's own (updated) tensor
                return o.cpu()
            return tuple(back(o) for o in out)
        return out
    return wrapper


def _on_device(fn):
    """CUDA-key implementation: the C-ABI acts on the CURRENT device (launch configuration caches, the CU count), so make
    the tensors' device current when it is not (a tensor on cuda:1 while cuda:0 is current)."""
    def wrapper(*args):
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args)
                break
        return fn(*args)
    return wrapper


def _impl(name, fn, mutates=lambda a, args: False):
    fn._mutates = mutates
    _DEF.impl(name, _on_device(fn), "CUDA")
    _DEF.impl(name, _staged(fn), "CPU")


# ------------------------------------------------------------------------------------------------------
# convert_weight_packed            /root/reference/bench_moe.py:26-27,43-44; test_gemm.py:24
# ------------------------------------------------------------------------------------------------------
_DEF.define("convert_weight_packed(Tensor weight) -> Tensor")


def _pack_supported(rows, cols, dtype):
    if dtype == torch.bfloat16:
        return rows % 32 == 0 and cols % 8 == 0
    return rows % 16 == 0 and cols % 64 == 0


def _packed_bits(is_vnni, shape1, shape2, dtype):
    """bit 0: w1 is in packed order, bit 1: w2 is (a weight is re-tiled iff its own shape allows it)."""
    if not is_vnni:
        return 0
    return (1 if _pack_supported(shape1[0], shape1[1], dtype) else 0) | (2 if _pack_supported(shape2[0], shape2[1], dtype) else 0)


def convert_weight_packed(weight):
    if weight.dtype == torch.uint8:
        # MX-fp4 nibble pairs [N, K/2] (/root/reference/test_mxfp4.py:160,183): the W4A16 kernel reads them row-major,
        # only the scales have a packed order (convert_scale_packed)
        return weight.contiguous().clone()
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


# Opt-in "a8" mode of the fp8 path (sglk.h: SGLK_MOE_FP8_ACT): fp8 activations on the block-scaled fp8 matrix cores.  NOT
# the reference's W8A16 numerics, never a default: switched on by SGLK_FP8_ACT=1 in the environment at import, or by
# set_fp8_activations(True).
_fp8_act = os.environ.get("SGLK_FP8_ACT", "0") not in ("", "0")
last_path = 0   # sglk_fused_experts' path_taken of the most recent call (measurement reports)


def set_fp8_activations(on):
    global _fp8_act
    _fp8_act = bool(on)


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
    topk_weights = _f32c(topk_weights)
    if topk_ids.dtype != torch.int32 or not topk_ids.is_contiguous():
        topk_ids = topk_ids.to(torch.int32).contiguous()
    if not w1.is_contiguous() or not w2.is_contiguous():
        w1, w2 = w1.contiguous(), w2.contiguous()
    bn, bk = 0, 0
    if int(method) == FP8_W8A16:
        if block_size is None or len(block_size) != 2:
            raise RuntimeError("fused_experts: fp8 needs block_size = [block_n, block_k]")
        bn, bk = int(block_size[0]), int(block_size[1])
    if int(method) != UNQUANT:
        if w1_scale is None or w2_scale is None:
            raise RuntimeError("fused_experts: quantised modes need w1_scale and w2_scale")
        w1_scale = _f32c(w1_scale)
        w2_scale = _f32c(w2_scale)

    out = hidden_states_c if (inplace and hidden_states_c is hidden_states) else torch.empty_like(hidden_states_c)
    L = _lib.lib()
    wtype = _WTYPE[wdtype]
    flags = _lib.MOE_FP8_ACT if (_fp8_act and int(method) == FP8_W8A16) else 0
    pk = _packed_bits(is_vnni, (2 * N, K), (K, N), wdtype)
    if pk == 0 and M * topk >= 64 * E and _pack_supported(2 * N, K, wdtype) and _pack_supported(K, N, wdtype):
        flags |= _lib.MOE_PACK_WEIGHTS      # row-major weights at prefill sizes: re-tile into the workspace (one pass over them)
    ws_bytes = L.sglk_fused_experts_workspace_bytes_ex(M, N, K, E, topk, wtype, flags)
    ws = _workspace(ws_bytes, hidden_states.device)
    # second stream for the tail tiles: only the batch sizes that have them (full 256-row tiles plus short tails)
    aux = _aux(hidden_states.device) if (int(method) == FP8_W8A16 and 160 * E <= M * topk < 640 * E) else None
    path = ctypes.c_int32(0)
    args = _lib.FusedExpertsArgs(
        hidden=hidden_states_c.data_ptr(), hidden_stride=hidden_states_c.stride(0),
        out=out.data_ptr(), out_stride=out.stride(0),
        w1=w1.data_ptr(), w2=w2.data_ptr(),
        w1_scale=w1_scale.data_ptr() if w1_scale is not None else None,
        w2_scale=w2_scale.data_ptr() if w2_scale is not None else None,
        topk_weights=topk_weights.data_ptr(), topk_ids=topk_ids.data_ptr(),
        M=M, N=N, K=K, E=E, topk=topk, wtype=wtype,
        packed=_packed_bits(is_vnni, (2 * N, K), (K, N), wdtype),
        block_n=bn, block_k=bk, workspace=ws.data_ptr(), workspace_bytes=ws_bytes, stage_timer=_stage_timer,
        aux_stream=aux.stream if aux else None,
        aux_events=(ctypes.c_void_p * 2)(aux.ev0, aux.ev1) if aux else (ctypes.c_void_p * 2)(),
        flags=flags, path_taken=ctypes.pointer(path))
    rc = L.sglk_fused_experts(ctypes.byref(args), _stream(hidden_states))
    _lib.check(rc, "fused_experts_cpu")
    global last_path
    last_path = path.value
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


# ------------------------------------------------------------------------------------------------------
# fused_moe_block      router -> routed experts (-> shared expert) in one call.  The reference harness
#                      makes the calls separately: grouped_topk_cpu + fused_experts_cpu (/root/reference/test_moe.py:57-92),
#                      shared_expert_cpu on the routed output (/root/reference/test_shared_experts.py:34-40,68)
# ------------------------------------------------------------------------------------------------------
_DEF.define(
    "fused_moe_block(Tensor(a!) hidden_states, Tensor router_logits, Tensor w1, Tensor w2, int topk, bool renormalize, "
    "int num_expert_group, int topk_group, Tensor? correction_bias, bool inplace, bool use_int8_w8a8, bool use_fp8_w8a16, "
    "Tensor? w1_scale, Tensor? w2_scale, int[]? block_size, bool is_vnni, Tensor? shared_w1, Tensor? shared_w2, "
    "Tensor? shared_w1_scale, Tensor? shared_w2_scale, float routed_scaling_factor) -> (Tensor, Tensor, Tensor)")

_GATING_TYPE = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}


def fused_moe_block(hidden_states, router_logits, w1, w2, topk, renormalize, num_expert_group, topk_group, correction_bias,
                    inplace, use_int8_w8a8, use_fp8_w8a16, w1_scale, w2_scale, block_size, is_vnni, shared_w1, shared_w2,
                    shared_w1_scale, shared_w2_scale, routed_scaling_factor):
    """-> (out [M,K] bf16, topk_weights [M,topk] f32, topk_ids [M,topk] i32)"""
    if use_int8_w8a8 and use_fp8_w8a16:
        raise RuntimeError("fused_moe_block: use_int8_w8a8 and use_fp8_w8a16 are mutually exclusive")
    if hidden_states.dim() != 2 or w1.dim() != 3 or w2.dim() != 3 or hidden_states.dtype != torch.bfloat16:
        raise RuntimeError("fused_moe_block: expect bf16 hidden [M,K], w1 [E,2N,K], w2 [E,K,N]")
    M, K = hidden_states.shape
    E, N2, K1 = w1.shape
    N = N2 // 2
    if K1 != K or tuple(w2.shape) != (E, K, N) or tuple(router_logits.shape) != (M, E):
        raise RuntimeError("fused_moe_block: shape mismatch")
    if router_logits.dtype not in _GATING_TYPE:
        raise RuntimeError(f"fused_moe_block: router_logits dtype {router_logits.dtype}")
    wdtype = torch.int8 if use_int8_w8a8 else (torch.float8_e4m3fn if use_fp8_w8a16 else torch.bfloat16)
    if w1.dtype != wdtype or w2.dtype != wdtype:
        raise RuntimeError(f"fused_moe_block: weights must be {wdtype} for this mode")
    hs = hidden_states if hidden_states.stride(1) == 1 else hidden_states.contiguous()
    gating = router_logits if router_logits.stride(1) == 1 else router_logits.contiguous()
    bias = None
    if correction_bias is not None:
        bias = correction_bias.to(gating.dtype).contiguous()
    w1, w2 = w1.contiguous(), w2.contiguous()
    bn = bk = 0
    if use_fp8_w8a16:
        if block_size is None or len(block_size) != 2:
            raise RuntimeError("fused_moe_block: fp8 needs block_size = [block_n, block_k]")
        bn, bk = int(block_size[0]), int(block_size[1])
    if wdtype != torch.bfloat16:
        if w1_scale is None or w2_scale is None:
            raise RuntimeError("fused_moe_block: quantised modes need w1_scale and w2_scale")
        w1_scale, w2_scale = _f32c(w1_scale), _f32c(w2_scale)
    shared_N = 0
    if shared_w1 is not None:
        if shared_w2 is None or shared_w1.dim() != 2 or shared_w1.shape[1] != K or shared_w1.dtype != wdtype:
            raise RuntimeError("fused_moe_block: expect shared_w1 [2*Ns,K], shared_w2 [K,Ns] of the experts' dtype")
        shared_N = shared_w1.shape[0] // 2
        if tuple(shared_w2.shape) != (K, shared_N):
            raise RuntimeError("fused_moe_block: shared_w2 shape")
        shared_w1, shared_w2 = shared_w1.contiguous(), shared_w2.contiguous()
        if wdtype != torch.bfloat16:
            if shared_w1_scale is None or shared_w2_scale is None:
                raise RuntimeError("fused_moe_block: the shared expert needs its scales")
            shared_w1_scale, shared_w2_scale = _f32c(shared_w1_scale), _f32c(shared_w2_scale)
    out = hs if (inplace and hs is hidden_states) else torch.empty_like(hs)
    tw = torch.empty(M, topk, dtype=torch.float32, device=hs.device)
    ids = torch.empty(M, topk, dtype=torch.int32, device=hs.device)
    L = _lib.lib()
    wtype = _WTYPE[wdtype]
    flags = _lib.MOE_FP8_ACT if (_fp8_act and use_fp8_w8a16) else 0
    ws_bytes = L.sglk_moe_block_workspace_bytes(M, N, K, E, topk, wtype, flags, shared_N)
    ws = _workspace(ws_bytes, hs.device)
    path = ctypes.c_int32(0)
    ex = _lib.FusedExpertsArgs(
        hidden=hs.data_ptr(), hidden_stride=hs.stride(0), out=out.data_ptr(), out_stride=out.stride(0),
        w1=w1.data_ptr(), w2=w2.data_ptr(),
        w1_scale=w1_scale.data_ptr() if w1_scale is not None else None,
        w2_scale=w2_scale.data_ptr() if w2_scale is not None else None,
        topk_weights=tw.data_ptr(), topk_ids=ids.data_ptr(), M=M, N=N, K=K, E=E, topk=topk, wtype=wtype,
        packed=_packed_bits(is_vnni, (2 * N, K), (K, N), wdtype), block_n=bn, block_k=bk,
        workspace=ws.data_ptr(), workspace_bytes=ws_bytes, stage_timer=_stage_timer,
        aux_stream=None, aux_events=(ctypes.c_void_p * 2)(), flags=flags, path_taken=ctypes.pointer(path))
    args = _lib.MoeBlockArgs(
        experts=ex, gating=gating.data_ptr(), gating_stride=gating.stride(0), gating_type=_GATING_TYPE[gating.dtype],
        correction_bias=bias.data_ptr() if bias is not None else None,
        renormalize=int(bool(renormalize)), num_expert_group=int(num_expert_group), topk_group=int(topk_group),
        shared_N=shared_N,
        shared_w1=shared_w1.data_ptr() if shared_N else None, shared_w2=shared_w2.data_ptr() if shared_N else None,
        shared_w1_scale=shared_w1_scale.data_ptr() if (shared_N and shared_w1_scale is not None) else None,
        shared_w2_scale=shared_w2_scale.data_ptr() if (shared_N and shared_w2_scale is not None) else None,
        shared_packed=_packed_bits(is_vnni, (2 * shared_N, K), (K, shared_N), wdtype) if shared_N else 0,
        routed_scaling_factor=float(routed_scaling_factor))
    _lib.check(L.sglk_moe_block(ctypes.byref(args), _stream(hs)), "fused_moe_block")
    global last_path
    last_path = path.value
    if inplace and out is not hidden_states:
        hidden_states.copy_(out)
        out = hidden_states
    return out, tw, ids


def _mut_block(a, args):
    return a is args[0] and bool(args[9])


_impl("fused_moe_block", fused_moe_block, _mut_block)
_impl("fused_experts_cpu", fused_experts_cpu, _mut_inplace)
_impl("fused_experts_cpu.method", fused_experts_cpu_method, _mut_inplace)


# ------------------------------------------------------------------------------------------------------
# shared_expert_cpu     14-arg: /root/reference/test_moe_fp8.py:87-88, test_moe_fp8_ext.py:60-61
#                       12-arg: /root/reference/test_shared_experts.py:68,78
# ------------------------------------------------------------------------------------------------------
_DEF.define(
    "shared_expert_cpu(Tensor(a!) hidden_states, Tensor w1, Tensor w2, Tensor fused_experts_out, "
    "float routed_scaling_factor, bool inplace, bool use_int8_w8a8, bool use_fp8_w8a16, Tensor? w1_scale, "
    "Tensor? w2_scale, int[]? block_size, Tensor? a1_scale, Tensor? a2_scale, bool is_vnni) -> Tensor")
_DEF.define(
    "shared_expert_cpu.v12(Tensor(a!) hidden_states, Tensor w1, Tensor w2, Tensor fused_experts_out, "
    "float routed_scaling_factor, bool inplace, bool use_int8_w8a8, bool use_fp8_w8a16, Tensor? w1_scale, "
    "Tensor? w2_scale, int[]? block_size, bool is_vnni) -> Tensor")


def shared_expert_cpu(hidden_states, w1, w2, fused_experts_out, routed_scaling_factor, inplace, use_int8_w8a8,
                      use_fp8_w8a16, w1_scale, w2_scale, block_size, a1_scale, a2_scale, is_vnni):
    if use_int8_w8a8 and use_fp8_w8a16:
        raise RuntimeError("shared_expert: use_int8_w8a8 and use_fp8_w8a16 are mutually exclusive")
    if a1_scale is not None or a2_scale is not None:
        raise RuntimeError("shared_expert: static activation scales are not supported")
    if hidden_states.dim() != 2 or w1.dim() != 2 or w2.dim() != 2 or hidden_states.dtype != torch.bfloat16:
        raise RuntimeError("shared_expert: expect bf16 hidden [M,K], w1 [2N,K], w2 [K,N]")
    M, K = hidden_states.shape
    N = w1.shape[0] // 2
    if tuple(w1.shape) != (2 * N, K) or tuple(w2.shape) != (K, N) or tuple(fused_experts_out.shape) != (M, K):
        raise RuntimeError("shared_expert: shape mismatch")
    wdtype = torch.int8 if use_int8_w8a8 else (torch.float8_e4m3fn if use_fp8_w8a16 else torch.bfloat16)
    if w1.dtype != wdtype or w2.dtype != wdtype:
        raise RuntimeError(f"shared_expert: weights must be {wdtype} for this mode")
    hs = hidden_states if hidden_states.stride(1) == 1 else hidden_states.contiguous()
    fo = fused_experts_out.to(torch.bfloat16)
    fo = fo if fo.stride(1) == 1 else fo.contiguous()
    w1, w2 = w1.contiguous(), w2.contiguous()
    bn = bk = 0
    if use_fp8_w8a16:
        if block_size is None or len(block_size) != 2:
            raise RuntimeError("shared_expert: fp8 needs block_size = [block_n, block_k]")
        bn, bk = int(block_size[0]), int(block_size[1])
    if wdtype != torch.bfloat16:
        if w1_scale is None or w2_scale is None:
            raise RuntimeError("shared_expert: quantised modes need w1_scale and w2_scale")
        w1_scale = w1_scale.to(torch.float32).contiguous()
        w2_scale = w2_scale.to(torch.float32).contiguous()
    out = hs if (inplace and hs is hidden_states) else torch.empty_like(hs)
    L = _lib.lib()
    wtype = _WTYPE[wdtype]
    pk = _packed_bits(is_vnni, (2 * N, K), (K, N), wdtype)
    ws_bytes = L.sglk_shared_expert_workspace_bytes_ex(M, N, K, wtype, pk)   # row-major weights at prefill sizes: + re-tiled copies
    ws = _workspace(ws_bytes, hs.device)
    args = _lib.SharedExpertArgs(
        hidden=hs.data_ptr(), hidden_stride=hs.stride(0), out=out.data_ptr(), out_stride=out.stride(0),
        w1=w1.data_ptr(), w2=w2.data_ptr(),
        w1_scale=w1_scale.data_ptr() if w1_scale is not None else None,
        w2_scale=w2_scale.data_ptr() if w2_scale is not None else None,
        fused_out=fo.data_ptr(), fused_out_stride=fo.stride(0), routed_scaling_factor=float(routed_scaling_factor),
        M=M, N=N, K=K, wtype=wtype, packed=pk,
        block_n=bn, block_k=bk, workspace=ws.data_ptr(), workspace_bytes=ws_bytes)
    _lib.check(L.sglk_shared_expert(ctypes.byref(args), _stream(hs)), "shared_expert_cpu")
    if inplace and out is not hidden_states:
        hidden_states.copy_(out)
        return hidden_states
    return out


def shared_expert_cpu_v12(hidden_states, w1, w2, fused_experts_out, routed_scaling_factor, inplace, use_int8_w8a8,
                          use_fp8_w8a16, w1_scale, w2_scale, block_size, is_vnni):
    return shared_expert_cpu(hidden_states, w1, w2, fused_experts_out, routed_scaling_factor, inplace, use_int8_w8a8,
                             use_fp8_w8a16, w1_scale, w2_scale, block_size, None, None, is_vnni)


_impl("shared_expert_cpu", shared_expert_cpu, _mut_inplace)
_impl("shared_expert_cpu.v12", shared_expert_cpu_v12, _mut_inplace)

# ------------------------------------------------------------------------------------------------------
# dense GEMMs: weight_packed_linear (/root/reference/test_gemm.py:22-25), fp8_scaled_mm_cpu (test_gemm_fp8.py:54-62),
# per_token_quant_int8_cpu / int8_scaled_mm_cpu / int8_scaled_mm_with_quant (test_gemm_int8.py:66-72)
# ------------------------------------------------------------------------------------------------------
_DEF.define("weight_packed_linear(Tensor x, Tensor weight, Tensor? bias, bool is_vnni) -> Tensor")
_DEF.define("fp8_scaled_mm_cpu(Tensor mat1, Tensor mat2, Tensor scales2, int[] block_size, Tensor? bias, "
            "ScalarType out_dtype, bool is_vnni) -> Tensor")
_DEF.define("per_token_quant_int8_cpu(Tensor A) -> (Tensor, Tensor)")
_DEF.define("int8_scaled_mm_cpu(Tensor mat1, Tensor mat2, Tensor scales1, Tensor scales2, Tensor? bias, "
            "ScalarType out_dtype, bool is_vnni) -> Tensor")
_DEF.define("int8_scaled_mm_with_quant(Tensor mat1, Tensor mat2, Tensor scales2, Tensor? bias, "
            "ScalarType out_dtype, bool is_vnni) -> Tensor")

_OUT_TYPE = {torch.bfloat16: _lib.OUT_BF16, torch.float16: _lib.OUT_F16, torch.float32: _lib.OUT_F32}


def _scaled_mm(x, w, w_scale, bias, out_dtype, is_vnni, block, x_scale=None):
    if x.dim() != 2 or w.dim() != 2 or x.shape[1] != w.shape[1]:
        raise RuntimeError(f"scaled_mm: expect x [M,K], w [N,K] (got {tuple(x.shape)}, {tuple(w.shape)})")
    if out_dtype not in _OUT_TYPE:
        raise RuntimeError(f"scaled_mm: unsupported out_dtype {out_dtype}")
    if w.dtype not in _WTYPE:
        raise RuntimeError(f"scaled_mm: unsupported weight dtype {w.dtype}")
    x_is_int8 = x.dtype == torch.int8
    if not x_is_int8 and x.dtype != torch.bfloat16:
        raise RuntimeError(f"scaled_mm: activations must be bfloat16 or int8 (got {x.dtype})")
    M, K = x.shape
    N = w.shape[0]
    if x.stride(1) != 1:
        x = x.contiguous()
    w = w.contiguous()
    if w_scale is not None:
        w_scale = w_scale.to(torch.float32).contiguous()
    if bias is not None:
        bias = bias.to(torch.float32).contiguous()
    if x_scale is not None:
        x_scale = x_scale.to(torch.float32).contiguous().view(-1)
    out = torch.empty(M, N, dtype=out_dtype, device=x.device)
    L = _lib.lib()
    wtype = _WTYPE[w.dtype]
    packed = 1 if (is_vnni and _pack_supported(N, K, w.dtype)) else 0
    ws_bytes = L.sglk_scaled_mm_workspace_bytes_ex(M, N, K, wtype, int(x_is_int8), packed)
    ws = _workspace(ws_bytes, x.device)
    args = _lib.ScaledMmArgs(
        x=x.data_ptr(), x_stride=x.stride(0), x_is_int8=int(x_is_int8),
        x_scale=x_scale.data_ptr() if x_scale is not None else None, w=w.data_ptr(),
        w_scale=w_scale.data_ptr() if w_scale is not None else None,
        bias=bias.data_ptr() if bias is not None else None, out=out.data_ptr(), out_stride=out.stride(0),
        out_type=_OUT_TYPE[out_dtype], M=M, N=N, K=K, wtype=wtype,
        packed=packed,
        block_n=int(block[0]) if block else 0, block_k=int(block[1]) if block else 0,
        workspace=ws.data_ptr(), workspace_bytes=ws_bytes)
    _lib.check(L.sglk_scaled_mm(ctypes.byref(args), _stream(x)), "scaled_mm")
    return out


def weight_packed_linear(x, weight, bias, is_vnni):
    return _scaled_mm(x, weight, None, bias, x.dtype, is_vnni, None)


def fp8_scaled_mm_cpu(mat1, mat2, scales2, block_size, bias, out_dtype, is_vnni):
    if len(block_size) != 2:
        raise RuntimeError("fp8_scaled_mm: block_size must be [block_n, block_k]")
    return _scaled_mm(mat1, mat2, scales2, bias, out_dtype, is_vnni, block_size)


def per_token_quant_int8_cpu(A):
    if A.dim() != 2 or A.dtype != torch.bfloat16:
        raise RuntimeError("per_token_quant_int8: expect a 2-D bfloat16 tensor")
    A = A if A.stride(1) == 1 else A.contiguous()
    M, K = A.shape
    q = torch.empty(M, K, dtype=torch.int8, device=A.device)
    s = torch.empty(M, dtype=torch.float32, device=A.device)
    _lib.check(_lib.lib().sglk_per_token_quant_int8(_ptr(A), A.stride(0), _ptr(q), K, _ptr(s), M, K, _stream(A)),
               "per_token_quant_int8_cpu")
    return q, s


def int8_scaled_mm_cpu(mat1, mat2, scales1, scales2, bias, out_dtype, is_vnni):
    return _scaled_mm(mat1, mat2, scales2, bias, out_dtype, is_vnni, None, x_scale=scales1)


def int8_scaled_mm_with_quant(mat1, mat2, scales2, bias, out_dtype, is_vnni):
    return _scaled_mm(mat1, mat2, scales2, bias, out_dtype, is_vnni, None)


_impl("weight_packed_linear", weight_packed_linear)
_impl("fp8_scaled_mm_cpu", fp8_scaled_mm_cpu)
_impl("per_token_quant_int8_cpu", per_token_quant_int8_cpu)
_impl("int8_scaled_mm_cpu", int8_scaled_mm_cpu)
_impl("int8_scaled_mm_with_quant", int8_scaled_mm_with_quant)


# ------------------------------------------------------------------------------------------------------
# qkv_proj_with_rope: MLA "absorbed" q/k/v projection (/root/reference/test_absorb.py:133-147,184-186).  Host logic
# only: the operator is the reference's sequence of GEMM / RMSNorm / per-head product / RoPE, each step one C-ABI call
# (sglk_scaled_mm, sglk_rmsnorm, sglk_bmm_heads, sglk_rope_gptj), same op order and rounding points as the oracle
# (native_torch / native_torch_int8, test_absorb.py:65-109).
_DEF.define("qkv_proj_with_rope(Tensor hidden_states, Tensor q_a_proj_weight, Tensor q_b_proj_weight, "
            "Tensor kv_a_proj_weight, Tensor w_kc, Tensor q_a_layernorm_weight, Tensor kv_a_layernorm_weight, "
            "Tensor positions, Tensor cos_sin_cache, float eps, bool use_int8_w8a8, bool use_fp8_w8a16, "
            "Tensor? q_a_proj_scale, Tensor? q_b_proj_scale, Tensor? kv_a_proj_scale, bool is_vnni, "
            "int[]? block_size) -> (Tensor, Tensor, Tensor)")


def qkv_proj_with_rope(hidden_states, q_a_proj_weight, q_b_proj_weight, kv_a_proj_weight, w_kc, q_a_layernorm_weight,
                       kv_a_layernorm_weight, positions, cos_sin_cache, eps, use_int8_w8a8, use_fp8_w8a16,
                       q_a_proj_scale, q_b_proj_scale, kv_a_proj_scale, is_vnni, block_size):
    hs = hidden_states
    if hs.dim() != 2 or hs.dtype != torch.bfloat16:
        raise RuntimeError("qkv_proj_with_rope: hidden_states must be a 2-D bfloat16 tensor")
    if w_kc.dim() != 3 or w_kc.dtype != torch.bfloat16:
        raise RuntimeError("qkv_proj_with_rope: w_kc must be [num_heads, kv_lora_rank, qk_nope_head_dim] bfloat16")
    if use_int8_w8a8 and use_fp8_w8a16:
        raise RuntimeError("qkv_proj_with_rope: use_int8_w8a8 and use_fp8_w8a16 are mutually exclusive")
    B = hs.shape[0]
    H, R, nope = w_kc.shape
    if q_b_proj_weight.shape[0] % H != 0:
        raise RuntimeError("qkv_proj_with_rope: q_b_proj rows must be num_heads * qk_head_dim")
    qk_head = q_b_proj_weight.shape[0] // H
    rope_dim = qk_head - nope
    if rope_dim <= 0 or rope_dim % 2 or kv_a_proj_weight.shape[0] != R + rope_dim or cos_sin_cache.shape[-1] != rope_dim:
        raise RuntimeError("qkv_proj_with_rope: inconsistent head dims")
    if cos_sin_cache.dtype != torch.bfloat16 or positions.dtype not in (torch.int64, torch.int32):
        raise RuntimeError("qkv_proj_with_rope: cos_sin_cache must be bfloat16 and positions int64/int32")
    L = _lib.lib()
    QL, hidden = q_a_proj_weight.shape
    if hs.shape[1] != hidden or q_b_proj_weight.shape[1] != QL or kv_a_proj_weight.shape[1] != hidden or \
            q_a_layernorm_weight.numel() != QL or kv_a_layernorm_weight.numel() != R:
        raise RuntimeError("qkv_proj_with_rope: weight shapes do not agree")
    wdt = q_a_proj_weight.dtype
    if q_b_proj_weight.dtype != wdt or kv_a_proj_weight.dtype != wdt or wdt not in _WTYPE:
        raise RuntimeError("qkv_proj_with_rope: the three projection weights must share one supported dtype")
    if use_int8_w8a8 != (wdt == torch.int8) or use_fp8_w8a16 != (wdt == torch.float8_e4m3fn):
        raise RuntimeError("qkv_proj_with_rope: use_int8_w8a8 / use_fp8_w8a16 do not match the weight dtype")
    scales = [None, None, None]
    if use_int8_w8a8 or use_fp8_w8a16:
        if q_a_proj_scale is None or q_b_proj_scale is None or kv_a_proj_scale is None:
            raise RuntimeError("qkv_proj_with_rope: quantised weights need the three weight scales")
        scales = [t.to(torch.float32).contiguous() for t in (q_a_proj_scale, q_b_proj_scale, kv_a_proj_scale)]
    if use_fp8_w8a16 and (block_size is None or len(block_size) != 2):
        raise RuntimeError("qkv_proj_with_rope: fp8 needs block_size [block_n, block_k]")
    hs = hs if hs.stride(1) == 1 else hs.contiguous()
    ws_ = [t.contiguous() for t in (q_a_proj_weight, q_b_proj_weight, kv_a_proj_weight)]
    wk = w_kc if w_kc.is_contiguous() else w_kc.contiguous()
    ln1, ln2 = q_a_layernorm_weight.contiguous(), kv_a_layernorm_weight.contiguous()
    if ln1.dtype != torch.bfloat16 or ln2.dtype != torch.bfloat16:
        raise RuntimeError("qkv_proj_with_rope: layernorm weights must be bfloat16")
    pos = positions if positions.is_contiguous() else positions.contiguous()
    cache = cos_sin_cache if cos_sin_cache.stride(-1) == 1 else cos_sin_cache.contiguous()
    q_input = torch.empty(B, H, R + rope_dim, dtype=torch.bfloat16, device=hs.device)
    k_input = torch.empty(B, 1, R + rope_dim, dtype=torch.bfloat16, device=hs.device)
    v_input = torch.empty(B, 1, R, dtype=torch.bfloat16, device=hs.device)
    wtype = _WTYPE[wdt]
    nbytes = L.sglk_qkv_proj_workspace_bytes(B, hidden, H, QL, R, nope, rope_dim, wtype)
    ws = _workspace(nbytes, hs.device)

    def packed(w):
        return 1 if (is_vnni and _pack_supported(w.shape[0], w.shape[1], wdt)) else 0

    args = _lib.QkvProjArgs(
        hidden=_ptr(hs), hidden_stride=hs.stride(0), B=B, hidden_size=hidden, q_a_w=_ptr(ws_[0]), q_b_w=_ptr(ws_[1]),
        kv_a_w=_ptr(ws_[2]), q_a_scale=_ptr(scales[0]) if scales[0] is not None else None,
        q_b_scale=_ptr(scales[1]) if scales[1] is not None else None,
        kv_a_scale=_ptr(scales[2]) if scales[2] is not None else None, wtype=wtype, packed_q_a=packed(ws_[0]),
        packed_q_b=packed(ws_[1]), packed_kv_a=packed(ws_[2]), block_n=int(block_size[0]) if use_fp8_w8a16 else 0,
        block_k=int(block_size[1]) if use_fp8_w8a16 else 0, w_kc=_ptr(wk), w_kc_packed=1 if is_vnni else 0, q_a_ln=_ptr(ln1),
        kv_a_ln=_ptr(ln2), eps=float(eps), positions=_ptr(pos), positions_is64=1 if pos.dtype == torch.int64 else 0,
        cos_sin_cache=_ptr(cache), cache_stride=cache.stride(0), H=H, q_lora=QL, kv_lora=R, nope=nope, rope=rope_dim,
        q_input=_ptr(q_input), q_stride_b=q_input.stride(0), q_stride_h=q_input.stride(1), k_input=_ptr(k_input),
        k_stride_b=k_input.stride(0), v_input=_ptr(v_input), v_stride_b=v_input.stride(0), workspace=_ptr(ws),
        workspace_bytes=nbytes)
    _lib.check(L.sglk_qkv_proj_with_rope(ctypes.byref(args), _stream(hs)), "qkv_proj_with_rope")
    return q_input, k_input, v_input


_impl("qkv_proj_with_rope", qkv_proj_with_rope)


# ------------------------------------------------------------------------------------------------------
# silu_and_mul_cpu: returning form /root/reference/bench_silu_and_mul.py:31; out-param form
#                   sgl_kernel.ops._kernels.silu_and_mul_cpu(out, x) /root/reference/test_activation.py:25
# rmsnorm_cpu / fused_add_rmsnorm_cpu: out-param / in-place, /root/reference/test_norm.py:44,56
# ------------------------------------------------------------------------------------------------------
_DEF.define("silu_and_mul_cpu(Tensor input) -> Tensor")
_DEF.define("silu_and_mul_cpu.out(Tensor(a!) out, Tensor input) -> ()")
_DEF.define("rmsnorm_cpu(Tensor(a!) output, Tensor input, Tensor weight, float eps) -> ()")
_DEF.define("fused_add_rmsnorm_cpu(Tensor(a!) input, Tensor(b!) residual, Tensor weight, float eps) -> ()")


def _is_f16(t, what):
    if t.dtype == torch.float16:
        return 1
    if t.dtype == torch.bfloat16:
        return 0
    raise RuntimeError(f"{what}: only bfloat16 / float16 are supported (got {t.dtype})")


def _rows2d(t, what):
    if t.stride(-1) != 1:
        raise RuntimeError(f"{what}: innermost dimension must be contiguous")
    if t.dim() == 1:
        return t.unsqueeze(0)
    if t.dim() == 2:
        return t
    if not t.is_contiguous():
        raise RuntimeError(f"{what}: tensors with more than 2 dims must be contiguous")
    return t.view(-1, t.shape[-1])


def silu_and_mul_out(out, input):
    f16 = _is_f16(input, "silu_and_mul")
    if out.dtype != input.dtype or input.shape[-1] % 2 or out.shape[-1] * 2 != input.shape[-1] \
            or out.shape[:-1] != input.shape[:-1]:
        raise RuntimeError("silu_and_mul: expect out [..., d] and input [..., 2d] of the same dtype")
    x2, o2 = _rows2d(input, "silu_and_mul"), _rows2d(out, "silu_and_mul")
    rc = _lib.lib().sglk_silu_and_mul(_ptr(x2), x2.stride(0), _ptr(o2), o2.stride(0), x2.shape[0], o2.shape[1], f16,
                                      _stream(input))
    _lib.check(rc, "silu_and_mul_cpu")


def silu_and_mul_cpu(input):
    out = torch.empty(input.shape[:-1] + (input.shape[-1] // 2,), dtype=input.dtype, device=input.device)
    silu_and_mul_out(out, input)
    return out


def rmsnorm_cpu(output, input, weight, eps):
    f16 = _is_f16(input, "rmsnorm")
    if output.dtype != input.dtype or weight.dtype != input.dtype or output.shape != input.shape \
            or weight.shape != input.shape[-1:]:
        raise RuntimeError("rmsnorm: output/input/weight dtype or shape mismatch")
    x2, o2 = _rows2d(input, "rmsnorm"), _rows2d(output, "rmsnorm")
    rc = _lib.lib().sglk_rmsnorm(_ptr(o2), o2.stride(0), _ptr(x2), x2.stride(0), _ptr(weight.contiguous()), x2.shape[0],
                                 x2.shape[1], float(eps), f16, _stream(input))
    _lib.check(rc, "rmsnorm_cpu")


def fused_add_rmsnorm_cpu(input, residual, weight, eps):
    f16 = _is_f16(input, "fused_add_rmsnorm")
    if residual.dtype != input.dtype or weight.dtype != input.dtype or residual.shape != input.shape \
            or weight.shape != input.shape[-1:]:
        raise RuntimeError("fused_add_rmsnorm: input/residual/weight dtype or shape mismatch")
    x2, r2 = _rows2d(input, "fused_add_rmsnorm"), _rows2d(residual, "fused_add_rmsnorm")
    rc = _lib.lib().sglk_fused_add_rmsnorm(_ptr(x2), x2.stride(0), _ptr(r2), r2.stride(0), _ptr(weight.contiguous()),
                                           x2.shape[0], x2.shape[1], float(eps), f16, _stream(input))
    _lib.check(rc, "fused_add_rmsnorm_cpu")


_impl("silu_and_mul_cpu", silu_and_mul_cpu)
_impl("silu_and_mul_cpu.out", silu_and_mul_out, lambda a, args: a is args[0])
_impl("rmsnorm_cpu", rmsnorm_cpu, lambda a, args: a is args[0])
_impl("fused_add_rmsnorm_cpu", fused_add_rmsnorm_cpu, lambda a, args: a is args[0] or a is args[1])

# ------------------------------------------------------------------------------------------------------
# grouped_topk_cpu: returning 9-arg form /root/reference/test_moe.py:61-70; out-param 8-arg form
#                   /root/reference/test_grouped_topk.py:61-69;  biased_grouped_topk_cpu out-param 9-arg form
#                   /root/reference/test_biased_grouped_topk.py:71-80
# ------------------------------------------------------------------------------------------------------
_DEF.define("grouped_topk_cpu(Tensor hidden_states, Tensor gating_output, int topk, bool renormalize, "
            "int num_expert_group, int topk_group, int num_fused_shared_experts, float? routed_scaling_factor, "
            "Tensor? num_token_non_padded) -> (Tensor, Tensor)")
_DEF.define("grouped_topk_cpu.out(Tensor(a!) topk_weights, Tensor(b!) topk_ids, Tensor hidden_states, "
            "Tensor gating_output, int topk, bool renormalize, int num_expert_group, int topk_group) -> ()")
_DEF.define("biased_grouped_topk_cpu(Tensor hidden_states, Tensor gating_output, Tensor correction_bias, int topk, "
            "bool renormalize, int num_expert_group, int topk_group, int num_fused_shared_experts, "
            "float? routed_scaling_factor, Tensor? num_token_non_padded) -> (Tensor, Tensor)")
_DEF.define("biased_grouped_topk_cpu.out(Tensor(a!) topk_weights, Tensor(b!) topk_ids, Tensor hidden_states, "
            "Tensor gating_output, Tensor correction_bias, int topk, bool renormalize, int num_expert_group, "
            "int topk_group) -> ()")

_GATE_TYPE = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}


def _grouped_topk(topk_weights, topk_ids, hidden_states, gating, bias, topk, renormalize, G, topk_group):
    if gating.dim() != 2 or hidden_states.shape[0] != gating.shape[0]:
        raise RuntimeError("grouped_topk: number of tokens mismatch")
    if gating.dtype not in _GATE_TYPE:
        raise RuntimeError(f"grouped_topk: unsupported gating dtype {gating.dtype}")
    M, E = gating.shape
    if tuple(topk_weights.shape) != (M, topk) or tuple(topk_ids.shape) != (M, topk) \
            or topk_weights.dtype != torch.float32 or topk_ids.dtype != torch.int32 \
            or not topk_weights.is_contiguous() or not topk_ids.is_contiguous():
        raise RuntimeError("grouped_topk: topk_weights must be contiguous f32 [M,topk], topk_ids contiguous i32 [M,topk]")
    g = gating if gating.stride(1) == 1 else gating.contiguous()
    b = None
    if bias is not None:
        b = bias.to(gating.dtype).contiguous()
        if tuple(b.shape) != (E,):
            raise RuntimeError("biased_grouped_topk: correction_bias must be [E]")
    rc = _lib.lib().sglk_grouped_topk(_ptr(g), g.stride(0), _GATE_TYPE[g.dtype], _ptr(b), _ptr(topk_weights),
                                      _ptr(topk_ids), M, E, int(topk), int(bool(renormalize)), int(G), int(topk_group),
                                      _stream(g))
    _lib.check(rc, "grouped_topk_cpu")


def _check_unsupported(num_fused_shared_experts, routed_scaling_factor, num_token_non_padded):
    if num_fused_shared_experts or routed_scaling_factor is not None or num_token_non_padded is not None:
        raise RuntimeError("grouped_topk: num_fused_shared_experts / routed_scaling_factor / num_token_non_padded "
                           "are not supported (the reference harness only passes 0 / None)")


def grouped_topk_cpu(hidden_states, gating_output, topk, renormalize, num_expert_group, topk_group,
                     num_fused_shared_experts, routed_scaling_factor, num_token_non_padded):
    _check_unsupported(num_fused_shared_experts, routed_scaling_factor, num_token_non_padded)
    M = gating_output.shape[0]
    w = torch.empty(M, topk, dtype=torch.float32, device=gating_output.device)
    ids = torch.empty(M, topk, dtype=torch.int32, device=gating_output.device)
    _grouped_topk(w, ids, hidden_states, gating_output, None, topk, renormalize, num_expert_group, topk_group)
    return w, ids


def grouped_topk_out(topk_weights, topk_ids, hidden_states, gating_output, topk, renormalize, num_expert_group,
                     topk_group):
    _grouped_topk(topk_weights, topk_ids, hidden_states, gating_output, None, topk, renormalize, num_expert_group,
                  topk_group)


def biased_grouped_topk_cpu(hidden_states, gating_output, correction_bias, topk, renormalize, num_expert_group,
                            topk_group, num_fused_shared_experts, routed_scaling_factor, num_token_non_padded):
    _check_unsupported(num_fused_shared_experts, routed_scaling_factor, num_token_non_padded)
    M = gating_output.shape[0]
    w = torch.empty(M, topk, dtype=torch.float32, device=gating_output.device)
    ids = torch.empty(M, topk, dtype=torch.int32, device=gating_output.device)
    _grouped_topk(w, ids, hidden_states, gating_output, correction_bias, topk, renormalize, num_expert_group, topk_group)
    return w, ids


def biased_grouped_topk_out(topk_weights, topk_ids, hidden_states, gating_output, correction_bias, topk, renormalize,
                            num_expert_group, topk_group):
    _grouped_topk(topk_weights, topk_ids, hidden_states, gating_output, correction_bias, topk, renormalize,
                  num_expert_group, topk_group)


_out2 = lambda a, args: a is args[0] or a is args[1]  # noqa: E731
_impl("grouped_topk_cpu", grouped_topk_cpu)
_impl("grouped_topk_cpu.out", grouped_topk_out, _out2)
_impl("biased_grouped_topk_cpu", biased_grouped_topk_cpu)
_impl("biased_grouped_topk_cpu.out", biased_grouped_topk_out, _out2)


# ------------------------------------------------------------------------------------------------------
# extend_attention_cpu (/root/reference/test_extend.py:168-182), decode_attention_cpu (/root/reference/test_mla.py:115-128)
# both write into caller-allocated outputs and return None
# ------------------------------------------------------------------------------------------------------
_DEF.define("extend_attention_cpu(Tensor q_extend, Tensor k_extend, Tensor v_extend, Tensor(a!) o_extend, "
            "Tensor k_buffer, Tensor v_buffer, Tensor req_to_tokens, Tensor b_req_idx, Tensor b_seq_len, "
            "Tensor b_seq_len_extend, Tensor b_start_loc_extend, int max_len_extend, float sm_scale, "
            "float logit_cap) -> ()")
_DEF.define("decode_attention_cpu(Tensor query, Tensor(a!) k_buffer, Tensor(b!) v_buffer, Tensor(c!) output, "
            "Tensor key, Tensor value, Tensor loc, Tensor(d!) attn_logits, Tensor req_to_token, Tensor b_req_idx, "
            "Tensor b_seq_len, float sm_scale, float logit_cap) -> ()")


def _thd(t, what):
    if t.dim() != 3 or t.dtype != torch.bfloat16 or t.stride(2) != 1:
        raise RuntimeError(f"{what}: expect a bfloat16 [tokens, heads, dim] tensor with a contiguous last dim")
    return t


def _s2(t):
    return (ctypes.c_int64 * 2)(t.stride(0), t.stride(1))


def _index2d(t, what):
    if t.dim() != 2 or t.dtype not in (torch.int32, torch.int64):
        raise RuntimeError(f"{what}: expect a 2-D int32/int64 tensor")
    return t if t.stride(1) == 1 else t.contiguous()


def extend_attention_cpu(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, req_to_tokens, b_req_idx, b_seq_len,
                         b_seq_len_extend, b_start_loc_extend, max_len_extend, sm_scale, logit_cap):
    for t, n in ((q_extend, "q_extend"), (k_extend, "k_extend"), (v_extend, "v_extend"), (o_extend, "o_extend"),
                 (k_buffer, "k_buffer"), (v_buffer, "v_buffer")):
        _thd(t, "extend_attention: " + n)
    T, HQ, D = q_extend.shape
    HKV, DV = k_extend.shape[1], v_extend.shape[2]
    if k_extend.shape[2] != D or k_buffer.shape[2] != D or v_buffer.shape[2] != DV or tuple(o_extend.shape) != (T, HQ, DV):
        raise RuntimeError("extend_attention: head-dim mismatch between q/k/v/o")
    B = b_seq_len.shape[0]
    rtt = _index2d(req_to_tokens, "extend_attention: req_to_tokens")
    req = b_req_idx.to(torch.int64).contiguous()
    seq = b_seq_len.to(torch.int64).contiguous()
    ext = b_seq_len_extend.to(torch.int32).contiguous()
    start = b_start_loc_extend.to(torch.int32).contiguous()
    args = _lib.ExtendAttentionArgs(
        q=q_extend.data_ptr(), k_extend=k_extend.data_ptr(), v_extend=v_extend.data_ptr(), k_buffer=k_buffer.data_ptr(),
        v_buffer=v_buffer.data_ptr(), o=o_extend.data_ptr(), q_stride=_s2(q_extend), k_extend_stride=_s2(k_extend),
        v_extend_stride=_s2(v_extend), k_buffer_stride=_s2(k_buffer), v_buffer_stride=_s2(v_buffer), o_stride=_s2(o_extend),
        req_to_tokens=rtt.data_ptr(), req_to_tokens_stride=rtt.stride(0), req_to_tokens_is64=int(rtt.dtype == torch.int64),
        b_req_idx=req.data_ptr(), b_seq_len=seq.data_ptr(), b_seq_len_extend=ext.data_ptr(),
        b_start_loc_extend=start.data_ptr(), B=B, HQ=HQ, HKV=HKV, HBUF=k_buffer.shape[1], D=D, DV=DV,
        max_len_extend=int(max_len_extend), sm_scale=float(sm_scale), logit_cap=float(logit_cap))
    _lib.check(_lib.lib().sglk_extend_attention(ctypes.byref(args), _stream(q_extend)), "extend_attention_cpu")


# MX-fp4 W4A16 GEMM: convert_scale_packed, mxfp4_scaled_mm_cpu   /root/reference/test_mxfp4.py:5-7,160-171,183-202
_DEF.define("convert_scale_packed(Tensor scale) -> Tensor")
_DEF.define("mxfp4_scaled_mm_cpu(Tensor x, Tensor weight, Tensor scale, Tensor? bias, bool is_vnni) -> Tensor")


def convert_scale_packed(scale):
    """E8M0 scales [N, K/32] -> the order the reference checks (/root/reference/test_mxfp4.py:186): [N/32][K/32][32],
    returned with the input's shape.  N not a multiple of 32: unchanged (the GEMM then reads them row-major)."""
    if scale.dim() != 2 or scale.dtype != torch.uint8:
        raise RuntimeError("convert_scale_packed: expect a 2-D uint8 [N, K/32] tensor of E8M0 scales")
    n, kb = scale.shape
    if n % 32 != 0:
        return scale.contiguous().clone()
    return scale.view(n // 32, 32, kb).transpose(1, 2).contiguous().view(n, kb)


def mxfp4_scaled_mm_cpu(x, weight, scale, bias, is_vnni):
    if x.dim() != 2 or x.dtype != torch.bfloat16 or x.stride(1) != 1:
        raise RuntimeError("mxfp4_scaled_mm_cpu: x must be a 2-D bfloat16 tensor with a contiguous last dim")
    if weight.dim() != 2 or weight.dtype != torch.uint8 or scale.dim() != 2 or scale.dtype != torch.uint8:
        raise RuntimeError("mxfp4_scaled_mm_cpu: weight [N, K/2] and scale [N, K/32] must be uint8")
    M, K = x.shape
    N = weight.shape[0]
    if weight.shape[1] * 2 != K or K % 32 != 0 or tuple(scale.shape) != (N, K // 32):
        raise RuntimeError(f"mxfp4_scaled_mm_cpu: x {tuple(x.shape)}, weight {tuple(weight.shape)}, scale "
                           f"{tuple(scale.shape)} do not describe [M, K] x [N, K/2] with one scale per 32 (K % 32 == 0)")
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != N):
        raise RuntimeError("mxfp4_scaled_mm_cpu: bias must be float32 [N]")
    w = weight.contiguous()
    sc = scale.contiguous()
    b = bias.contiguous() if bias is not None else None
    out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
    L = _lib.lib()
    ws = _workspace(L.sglk_mxfp4_workspace_bytes(M, N, K), x.device)
    _lib.check(L.sglk_mxfp4_scaled_mm(_ptr(x), x.stride(0), _ptr(w), _ptr(sc), 1 if (is_vnni and N % 32 == 0) else 0,
                                      _ptr(b) if b is not None else None, _ptr(out), out.stride(0), M, N, K, _ptr(ws),
                                      ws.numel(), _stream(x)), "mxfp4_scaled_mm_cpu")
    return out


_impl("convert_scale_packed", convert_scale_packed)
_impl("mxfp4_scaled_mm_cpu", mxfp4_scaled_mm_cpu)


# bmm_cpu: /root/reference/test_bmm_fp8.py:38-39,67,73 -- out[b] = mat1[b] @ mat2[b]^T, bf16, strided out / mat1 views
_DEF.define("bmm_cpu(Tensor(a!) out, Tensor mat1, Tensor mat2, bool is_vnni, Tensor? scale) -> ()")


def bmm_cpu(out, mat1, mat2, is_vnni, scale):
    if scale is not None:
        raise RuntimeError("bmm_cpu: scaled (fp8) mat2 is not supported; the reference harness passes scale=None")
    for t, n in ((out, "out"), (mat1, "mat1"), (mat2, "mat2")):
        if t.dim() != 3 or t.dtype != torch.bfloat16:
            raise RuntimeError(f"bmm_cpu: {n} must be a 3-D bfloat16 tensor")
    Bn, M, K = mat1.shape
    N = mat2.shape[1]
    if mat2.shape[0] != Bn or mat2.shape[2] != K or tuple(out.shape) != (Bn, M, N):
        raise RuntimeError(f"bmm_cpu: shapes out {tuple(out.shape)}, mat1 {tuple(mat1.shape)}, mat2 [B, N, K] "
                           f"{tuple(mat2.shape)} do not agree")
    if mat1.stride(2) != 1 or out.stride(2) != 1 or not mat2.is_contiguous():
        raise RuntimeError("bmm_cpu: out / mat1 need a contiguous last dim, mat2 must be contiguous [B, N, K]")
    # kernel naming: "heads" = the bmm batch, "rows" = M (sglk.h: out[b][h][oc] = sum_ic x[b][h][ic] * w[h][oc][ic])
    packed = 1 if (is_vnni and _pack_supported(N, K, mat2.dtype)) else 0
    _lib.check(_lib.lib().sglk_bmm_heads(_ptr(mat1), mat1.stride(1), mat1.stride(0), _ptr(mat2), packed, _ptr(out),
                                         out.stride(1), out.stride(0), M, Bn, N, K, _stream(out)), "bmm_cpu")


_impl("bmm_cpu", bmm_cpu, lambda a, args: a is args[0])


# flash_attn_varlen_func: /root/reference/test_flash_attn_varlen.py:100-108,148-153
_DEF.define("flash_attn_varlen_func(Tensor q, Tensor k, Tensor v, Tensor cu_seqlens_q, Tensor cu_seqlens_k, "
            "int max_seqlen_q, int max_seqlen_k, bool causal) -> Tensor")


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal):
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _thd(t, "flash_attn_varlen_func: " + n)
    Tq, HQ, D = q.shape
    Tk, HKV, DV = v.shape
    if k.shape[0] != Tk or k.shape[1] != HKV or k.shape[2] != D:
        raise RuntimeError("flash_attn_varlen_func: k must be [total_k, num_heads_kv, head_dim] like q / v")
    if cu_seqlens_q.dim() != 1 or cu_seqlens_q.shape != cu_seqlens_k.shape or cu_seqlens_q.numel() < 1:
        raise RuntimeError("flash_attn_varlen_func: cu_seqlens_q / cu_seqlens_k must be 1-D with batch + 1 entries")
    cq = cu_seqlens_q.to(torch.int32).contiguous()
    ck = cu_seqlens_k.to(torch.int32).contiguous()
    out = torch.empty(Tq, HQ, DV, dtype=q.dtype, device=q.device)
    args = _lib.FlashAttnVarlenArgs(
        q=q.data_ptr(), q_stride=_s2(q), k=k.data_ptr(), k_stride=_s2(k), v=v.data_ptr(), v_stride=_s2(v), o=out.data_ptr(),
        o_stride=_s2(out), cu_seqlens_q=cq.data_ptr(), cu_seqlens_k=ck.data_ptr(), B=cq.numel() - 1,
        max_seqlen_q=int(max_seqlen_q), HQ=HQ, HKV=HKV, D=D, DV=DV, causal=int(bool(causal)), sm_scale=1.0 / D ** 0.5)
    _lib.check(_lib.lib().sglk_flash_attn_varlen(ctypes.byref(args), _stream(q)), "flash_attn_varlen_func")
    return out


_impl("flash_attn_varlen_func", flash_attn_varlen_func)


def decode_attention_cpu(query, k_buffer, v_buffer, output, key, value, loc, attn_logits, req_to_token, b_req_idx,
                         b_seq_len, sm_scale, logit_cap):
    for t, n in ((query, "query"), (k_buffer, "k_buffer"), (v_buffer, "v_buffer"), (output, "output"), (key, "key"),
                 (value, "value")):
        _thd(t, "decode_attention: " + n)
    B, HQ, D = query.shape
    HKV, DV = k_buffer.shape[1], v_buffer.shape[2]
    if tuple(output.shape) != (B, HQ, DV) or tuple(key.shape) != (B, HKV, D) or tuple(value.shape) != (B, HKV, DV):
        raise RuntimeError("decode_attention: shape mismatch")
    if attn_logits.dim() != 4 or attn_logits.dtype != torch.float32 or not attn_logits.is_contiguous() \
            or attn_logits.shape[0] != B or attn_logits.shape[1] != HQ or attn_logits.shape[3] != DV + 1:
        raise RuntimeError("decode_attention: attn_logits must be contiguous f32 [B, HQ, splits, DV+1]")
    if loc.dtype not in (torch.int32, torch.int64):
        raise RuntimeError("decode_attention: loc must be int32/int64")
    rtt = _index2d(req_to_token, "decode_attention: req_to_token")
    req = b_req_idx.to(torch.int64).contiguous()
    seq = b_seq_len.to(torch.int64).contiguous()
    loc = loc.contiguous()
    args = _lib.DecodeAttentionArgs(
        q=query.data_ptr(), k_buffer=k_buffer.data_ptr(), v_buffer=v_buffer.data_ptr(), o=output.data_ptr(),
        key=key.data_ptr(), value=value.data_ptr(), q_stride=_s2(query), k_buffer_stride=_s2(k_buffer),
        v_buffer_stride=_s2(v_buffer), o_stride=_s2(output), key_stride=_s2(key), value_stride=_s2(value),
        loc=loc.data_ptr(), loc_is64=int(loc.dtype == torch.int64), attn_logits=attn_logits.data_ptr(),
        req_to_token=rtt.data_ptr(), req_to_token_stride=rtt.stride(0), req_to_token_is64=int(rtt.dtype == torch.int64),
        b_req_idx=req.data_ptr(), b_seq_len=seq.data_ptr(), B=B, HQ=HQ, HKV=HKV, D=D, DV=DV,
        splits=attn_logits.shape[2], sm_scale=float(sm_scale), logit_cap=float(logit_cap))
    _lib.check(_lib.lib().sglk_decode_attention(ctypes.byref(args), _stream(query)), "decode_attention_cpu")


_impl("extend_attention_cpu", extend_attention_cpu, lambda a, args: a is args[3])
_impl("decode_attention_cpu", decode_attention_cpu, lambda a, args: a is args[1] or a is args[2] or a is args[3] or a is args[7])
