"""ctypes view of libsglk.so — the C-ABI declared in include/sglk.h.  No compute happens in Python."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGLK_LIB_PATH") or os.path.join(_HERE, "libsglk.so")   # override: developer A/B builds

W_BF16, W_FP8_E4M3, W_INT8 = 0, 1, 2
MOE_FP8_ACT = 1                      # sglk_fused_experts_args.flags
MOE_PACK_WEIGHTS = 2                 # row-major weights: re-tile them into the workspace, run the packed kernels
PATH_TILE_MASK, PATH_FP8_ACT, PATH_TAILS_SPLIT, PATH_TAILS_AUX, PATH_PERSIST_G1, PATH_PERSIST_G2 = (
    0x3ff, 0x1000, 0x2000, 0x4000, 0x8000, 0x10000)
PATH_ROUTE_ALIGN, PATH_SHARED_FOLDED, PATH_SPLIT, PATH_INLINE_ALIGN = 0x20000, 0x40000, 0x80000, 0x100000


class FusedExpertsArgs(ctypes.Structure):
    """Mirror of `sglk_fused_experts_args` (include/sglk.h)."""
    _fields_ = [
        ("hidden", ctypes.c_void_p), ("hidden_stride", ctypes.c_int64),
        ("out", ctypes.c_void_p), ("out_stride", ctypes.c_int64),
        ("w1", ctypes.c_void_p), ("w2", ctypes.c_void_p),
        ("w1_scale", ctypes.c_void_p), ("w2_scale", ctypes.c_void_p),
        ("topk_weights", ctypes.c_void_p), ("topk_ids", ctypes.c_void_p),
        ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("E", ctypes.c_int32),
        ("topk", ctypes.c_int32), ("wtype", ctypes.c_int32), ("packed", ctypes.c_int32),
        ("block_n", ctypes.c_int32), ("block_k", ctypes.c_int32),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
        ("stage_timer", ctypes.c_void_p),
        ("aux_stream", ctypes.c_void_p), ("aux_events", ctypes.c_void_p * 2),
        ("flags", ctypes.c_int32), ("path_taken", ctypes.POINTER(ctypes.c_int32)),
    ]


class MoeBlockArgs(ctypes.Structure):
    """Mirror of `sglk_moe_block_args`."""
    _fields_ = [
        ("experts", FusedExpertsArgs),
        ("gating", ctypes.c_void_p), ("gating_stride", ctypes.c_int64), ("gating_type", ctypes.c_int32),
        ("correction_bias", ctypes.c_void_p),
        ("renormalize", ctypes.c_int32), ("num_expert_group", ctypes.c_int32), ("topk_group", ctypes.c_int32),
        ("shared_N", ctypes.c_int32),
        ("shared_w1", ctypes.c_void_p), ("shared_w2", ctypes.c_void_p),
        ("shared_w1_scale", ctypes.c_void_p), ("shared_w2_scale", ctypes.c_void_p),
        ("shared_packed", ctypes.c_int32), ("routed_scaling_factor", ctypes.c_float),
    ]


class SharedExpertArgs(ctypes.Structure):
    """Mirror of `sglk_shared_expert_args`."""
    _fields_ = [
        ("hidden", ctypes.c_void_p), ("hidden_stride", ctypes.c_int64),
        ("out", ctypes.c_void_p), ("out_stride", ctypes.c_int64),
        ("w1", ctypes.c_void_p), ("w2", ctypes.c_void_p),
        ("w1_scale", ctypes.c_void_p), ("w2_scale", ctypes.c_void_p),
        ("fused_out", ctypes.c_void_p), ("fused_out_stride", ctypes.c_int64),
        ("routed_scaling_factor", ctypes.c_float),
        ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
        ("wtype", ctypes.c_int32), ("packed", ctypes.c_int32),
        ("block_n", ctypes.c_int32), ("block_k", ctypes.c_int32),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
    ]


class ScaledMmArgs(ctypes.Structure):
    """Mirror of `sglk_scaled_mm_args`."""
    _fields_ = [
        ("x", ctypes.c_void_p), ("x_stride", ctypes.c_int64), ("x_is_int8", ctypes.c_int32),
        ("x_scale", ctypes.c_void_p), ("w", ctypes.c_void_p), ("w_scale", ctypes.c_void_p),
        ("bias", ctypes.c_void_p), ("out", ctypes.c_void_p), ("out_stride", ctypes.c_int64),
        ("out_type", ctypes.c_int32), ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
        ("wtype", ctypes.c_int32), ("packed", ctypes.c_int32), ("block_n", ctypes.c_int32), ("block_k", ctypes.c_int32),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
    ]


class QkvProjArgs(ctypes.Structure):
    """Mirror of `sglk_qkv_proj_args`."""
    _fields_ = [
        ("hidden", ctypes.c_void_p), ("hidden_stride", ctypes.c_int64), ("B", ctypes.c_int32), ("hidden_size", ctypes.c_int32),
        ("q_a_w", ctypes.c_void_p), ("q_b_w", ctypes.c_void_p), ("kv_a_w", ctypes.c_void_p),
        ("q_a_scale", ctypes.c_void_p), ("q_b_scale", ctypes.c_void_p), ("kv_a_scale", ctypes.c_void_p),
        ("wtype", ctypes.c_int32), ("packed_q_a", ctypes.c_int32), ("packed_q_b", ctypes.c_int32), ("packed_kv_a", ctypes.c_int32),
        ("block_n", ctypes.c_int32), ("block_k", ctypes.c_int32),
        ("w_kc", ctypes.c_void_p), ("w_kc_packed", ctypes.c_int32),
        ("q_a_ln", ctypes.c_void_p), ("kv_a_ln", ctypes.c_void_p), ("eps", ctypes.c_float),
        ("positions", ctypes.c_void_p), ("positions_is64", ctypes.c_int32),
        ("cos_sin_cache", ctypes.c_void_p), ("cache_stride", ctypes.c_int64),
        ("H", ctypes.c_int32), ("q_lora", ctypes.c_int32), ("kv_lora", ctypes.c_int32), ("nope", ctypes.c_int32), ("rope", ctypes.c_int32),
        ("q_input", ctypes.c_void_p), ("q_stride_b", ctypes.c_int64), ("q_stride_h", ctypes.c_int64),
        ("k_input", ctypes.c_void_p), ("k_stride_b", ctypes.c_int64),
        ("v_input", ctypes.c_void_p), ("v_stride_b", ctypes.c_int64),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
    ]


class ExtendAttentionArgs(ctypes.Structure):
    """Mirror of `sglk_extend_attention_args`."""
    _fields_ = [
        ("q", ctypes.c_void_p), ("k_extend", ctypes.c_void_p), ("v_extend", ctypes.c_void_p),
        ("k_buffer", ctypes.c_void_p), ("v_buffer", ctypes.c_void_p), ("o", ctypes.c_void_p),
        ("q_stride", ctypes.c_int64 * 2), ("k_extend_stride", ctypes.c_int64 * 2), ("v_extend_stride", ctypes.c_int64 * 2),
        ("k_buffer_stride", ctypes.c_int64 * 2), ("v_buffer_stride", ctypes.c_int64 * 2), ("o_stride", ctypes.c_int64 * 2),
        ("req_to_tokens", ctypes.c_void_p), ("req_to_tokens_stride", ctypes.c_int64), ("req_to_tokens_is64", ctypes.c_int32),
        ("b_req_idx", ctypes.c_void_p), ("b_seq_len", ctypes.c_void_p), ("b_seq_len_extend", ctypes.c_void_p),
        ("b_start_loc_extend", ctypes.c_void_p),
        ("B", ctypes.c_int32), ("HQ", ctypes.c_int32), ("HKV", ctypes.c_int32), ("HBUF", ctypes.c_int32),
        ("D", ctypes.c_int32), ("DV", ctypes.c_int32), ("max_len_extend", ctypes.c_int32),
        ("sm_scale", ctypes.c_float), ("logit_cap", ctypes.c_float),
    ]


class FlashAttnVarlenArgs(ctypes.Structure):
    """Mirror of `sglk_flash_attn_varlen_args`."""
    _fields_ = [
        ("q", ctypes.c_void_p), ("q_stride", ctypes.c_int64 * 2),
        ("k", ctypes.c_void_p), ("k_stride", ctypes.c_int64 * 2),
        ("v", ctypes.c_void_p), ("v_stride", ctypes.c_int64 * 2),
        ("o", ctypes.c_void_p), ("o_stride", ctypes.c_int64 * 2),
        ("cu_seqlens_q", ctypes.c_void_p), ("cu_seqlens_k", ctypes.c_void_p),
        ("B", ctypes.c_int32), ("max_seqlen_q", ctypes.c_int32), ("HQ", ctypes.c_int32), ("HKV", ctypes.c_int32),
        ("D", ctypes.c_int32), ("DV", ctypes.c_int32), ("causal", ctypes.c_int32), ("sm_scale", ctypes.c_float),
    ]


class DecodeAttentionArgs(ctypes.Structure):
    """Mirror of `sglk_decode_attention_args`."""
    _fields_ = [
        ("q", ctypes.c_void_p), ("k_buffer", ctypes.c_void_p), ("v_buffer", ctypes.c_void_p), ("o", ctypes.c_void_p),
        ("key", ctypes.c_void_p), ("value", ctypes.c_void_p),
        ("q_stride", ctypes.c_int64 * 2), ("k_buffer_stride", ctypes.c_int64 * 2), ("v_buffer_stride", ctypes.c_int64 * 2),
        ("o_stride", ctypes.c_int64 * 2), ("key_stride", ctypes.c_int64 * 2), ("value_stride", ctypes.c_int64 * 2),
        ("loc", ctypes.c_void_p), ("loc_is64", ctypes.c_int32), ("attn_logits", ctypes.c_void_p),
        ("req_to_token", ctypes.c_void_p), ("req_to_token_stride", ctypes.c_int64), ("req_to_token_is64", ctypes.c_int32),
        ("b_req_idx", ctypes.c_void_p), ("b_seq_len", ctypes.c_void_p),
        ("B", ctypes.c_int32), ("HQ", ctypes.c_int32), ("HKV", ctypes.c_int32), ("D", ctypes.c_int32),
        ("DV", ctypes.c_int32), ("splits", ctypes.c_int32), ("sm_scale", ctypes.c_float), ("logit_cap", ctypes.c_float),
    ]


OUT_BF16, OUT_F16, OUT_F32 = 0, 1, 2

# symbol -> (restype, argtypes); every symbol include/sglk.h declares must be listed here
# (tests/test_cabi_symbols.py checks the two against each other)
_SIGNATURES = {
    "sglk_version": (ctypes.c_int, []),
    "sglk_last_error": (ctypes.c_char_p, []),
    "sglk_device_cu_count": (ctypes.c_int, [ctypes.c_int]),
    "sglk_reload_env": (None, []),
    "sglk_aux_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p)] * 3),
    "sglk_aux_destroy": (None, [ctypes.c_void_p] * 3),
    "sglk_pack_weight": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                         ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "sglk_unpack_weight": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                           ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "sglk_fused_experts_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int32] * 6),
    "sglk_fused_experts_workspace_bytes_ex": (ctypes.c_size_t, [ctypes.c_int32] * 7),
    "sglk_fused_experts": (ctypes.c_int, [ctypes.POINTER(FusedExpertsArgs), ctypes.c_void_p]),
    "sglk_moe_block_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int32] * 8),
    "sglk_moe_block": (ctypes.c_int, [ctypes.POINTER(MoeBlockArgs), ctypes.c_void_p]),
    "sglk_quant_fp8_block128": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                               ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p]),
    "sglk_split_fp8_block128": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                               ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p]),
    "sglk_moe_align_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int32] * 3),
    "sglk_moe_max_tiles": (ctypes.c_int32, [ctypes.c_int32] * 4),
    "sglk_moe_align": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "sglk_shared_expert_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int32] * 4),
    "sglk_shared_expert_workspace_bytes_ex": (ctypes.c_size_t, [ctypes.c_int32] * 5),
    "sglk_shared_expert": (ctypes.c_int, [ctypes.POINTER(SharedExpertArgs), ctypes.c_void_p]),
    "sglk_scaled_mm_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int32] * 5),
    "sglk_scaled_mm": (ctypes.c_int, [ctypes.POINTER(ScaledMmArgs), ctypes.c_void_p]),
    "sglk_per_token_quant_int8": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                                  ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p]),
    "sglk_per_token_quant_int8_floor": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                                        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float,
                                                        ctypes.c_void_p]),
    "sglk_comm_alloc": (ctypes.c_int, [ctypes.c_size_t, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]),
    "sglk_comm_free": (None, [ctypes.c_void_p]),
    "sglk_ipc_export": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p]),
    "sglk_ipc_open": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]),
    "sglk_ipc_close": (ctypes.c_int, [ctypes.c_void_p]),
    "sglk_allreduce_sum_bf16": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32,
                                               ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                               ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "sglk_scaled_mm_workspace_bytes_ex": (ctypes.c_size_t, [ctypes.c_int32] * 6),
    "sglk_qkv_proj_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int32] * 8),
    "sglk_qkv_proj_with_rope": (ctypes.c_int, [ctypes.POINTER(QkvProjArgs), ctypes.c_void_p]),
    "sglk_mxfp4_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "sglk_ep_plan": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sglk_ep_pack": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                    ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "sglk_ep_reduce_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                           ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "sglk_mxfp4_scaled_mm": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.c_int32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "sglk_bmm_heads": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32,
                                       ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                       ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "sglk_rope_gptj": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                       ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                       ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "sglk_silu_and_mul": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "sglk_rmsnorm": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                     ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_int32, ctypes.c_void_p]),
    "sglk_fused_add_rmsnorm": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                               ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float,
                                               ctypes.c_int32, ctypes.c_void_p]),
    "sglk_grouped_topk": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_void_p]),
    "sglk_extend_attention": (ctypes.c_int, [ctypes.POINTER(ExtendAttentionArgs), ctypes.c_void_p]),
    "sglk_flash_attn_varlen": (ctypes.c_int, [ctypes.POINTER(FlashAttnVarlenArgs), ctypes.c_void_p]),
    "sglk_decode_attention": (ctypes.c_int, [ctypes.POINTER(DecodeAttentionArgs), ctypes.c_void_p]),
    "sglk_stage_timer_create": (ctypes.c_void_p, [ctypes.c_int32]),
    "sglk_stage_timer_destroy": (None, [ctypes.c_void_p]),
    "sglk_stage_timer_reset": (None, [ctypes.c_void_p]),
    "sglk_stage_timer_read": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float),
                                              ctypes.POINTER(ctypes.c_int32)]),
}
NUM_STAGES = 4
STAGE_NAMES = ("align", "gemm1_gate_up_silu", "gemm2_down", "combine")

_lib = None


def lib():
    """The loaded library.  Fails loudly when the HIP extension has not been built: there is NO fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"sgl_kernel: {LIB_PATH} is missing — build it with `python sgl-cpu-tests_amd/build.py` "
                "(hipcc, gfx950). This package has no CPU or PyTorch fallback.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            f = getattr(l, name)
            f.restype, f.argtypes = res, args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().sglk_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"sgl_kernel::{what} failed ({rc}): {msg}")
