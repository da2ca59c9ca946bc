// qkv_proj_with_rope (/root/reference/test_absorb.py:133-147,184-186; oracles native_torch :65-87, native_torch_int8 :89-131)
// as ONE C-ABI call: the MLA "absorbed" projection
//     q = rmsnorm(hidden . q_a^T) . q_b^T        latent = hidden . kv_a^T
//     q_input[b, h] = [ q_nope[b, h] . w_kc[h] | rope(q_pe[b, h]) ]
//     v_input[b] = rmsnorm(latent[b, :R]);  k_input[b] = [ v_input[b] | rope(k_pe[b]) ]
// The steps are this library's own entry points (sglk_scaled_mm for the three projections in bf16 / fp8 W8A16 / int8 W8A8,
// sglk_rmsnorm, sglk_bmm_heads, sglk_rope_gptj, sglk_per_token_quant_int8_floor), in the oracle's order with the oracle's rounding
// points (every intermediate is a bf16 tensor, as in native_torch) -- the call exists so that a decode step costs one host call and
// one workspace instead of eleven calls and a dozen allocations (ten launches); intermediates live in the caller's workspace.
#include "moe_internal.h"
#include "sglk.h"

namespace sglk {
namespace {
struct QkvWs {
    size_t qa, qn, q2, latent, xq, xs, mm, total, mm_bytes;
};
QkvWs plan_qkv(int B, int hidden, int H, int q_lora, int R, int nope, int rope, int wtype) {
    QkvWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += align_up(bytes > 0 ? bytes : 1, 256);
        return o;
    };
    const int qk = nope + rope;
    w.qa = take((size_t)B * q_lora * 2);
    w.qn = take((size_t)B * q_lora * 2);
    w.q2 = take((size_t)B * H * qk * 2);
    w.latent = take((size_t)B * (R + rope) * 2);
    const int kmax = hidden > q_lora ? hidden : q_lora;
    w.xq = take(wtype == SGLK_W_INT8 ? (size_t)B * kmax : 0);
    w.xs = take(wtype == SGLK_W_INT8 ? (size_t)B * 4 : 0);
    const int i8 = wtype == SGLK_W_INT8 ? 1 : 0;
    size_t m = sglk_scaled_mm_workspace_bytes(B, q_lora, hidden, wtype, i8);
    const size_t m2 = sglk_scaled_mm_workspace_bytes(B, H * qk, q_lora, wtype, i8);
    const size_t m3 = sglk_scaled_mm_workspace_bytes(B, R + rope, hidden, wtype, i8);
    m = m2 > m ? m2 : m;
    m = m3 > m ? m3 : m;
    w.mm_bytes = m;
    w.mm = take(m);
    w.total = off;
    return w;
}
}  // namespace
}  // namespace sglk

using namespace sglk;

extern "C" size_t sglk_qkv_proj_workspace_bytes(int32_t B, int32_t hidden, int32_t H, int32_t q_lora, int32_t kv_lora, int32_t nope,
                                                int32_t rope, int32_t wtype) {
    if (B < 0 || hidden <= 0 || H <= 0 || q_lora <= 0 || kv_lora <= 0 || nope <= 0 || rope <= 0) return 0;
    return plan_qkv(B, hidden, H, q_lora, kv_lora, nope, rope, wtype).total;
}

extern "C" int sglk_qkv_proj_with_rope(const sglk_qkv_proj_args* a, void* stream) {
    SGLK_REQUIRE(a, SGLK_ERR_INVALID, "qkv_proj_with_rope: null args");
    const int B = a->B, hidden = a->hidden_size, H = a->H, QL = a->q_lora, R = a->kv_lora, nope = a->nope, rope = a->rope;
    SGLK_REQUIRE(B >= 0 && hidden > 0 && H > 0 && QL > 0 && R > 0 && nope > 0 && rope > 0 && rope % 2 == 0, SGLK_ERR_INVALID,
                 "qkv_proj_with_rope: bad sizes");
    SGLK_REQUIRE(a->wtype == SGLK_W_BF16 || a->wtype == SGLK_W_FP8_E4M3 || a->wtype == SGLK_W_INT8, SGLK_ERR_INVALID,
                 "qkv_proj_with_rope: weight type %d", a->wtype);
    if (B == 0) return SGLK_OK;
    SGLK_REQUIRE(a->hidden && a->q_a_w && a->q_b_w && a->kv_a_w && a->w_kc && a->q_a_ln && a->kv_a_ln && a->positions &&
                     a->cos_sin_cache && a->q_input && a->k_input && a->v_input && a->workspace,
                 SGLK_ERR_INVALID, "qkv_proj_with_rope: null pointer");
    SGLK_REQUIRE(a->wtype == SGLK_W_BF16 || (a->q_a_scale && a->q_b_scale && a->kv_a_scale), SGLK_ERR_INVALID,
                 "qkv_proj_with_rope: quantised weights need their three scale tensors");
    const QkvWs w = plan_qkv(B, hidden, H, QL, R, nope, rope, a->wtype);
    SGLK_REQUIRE(a->workspace_bytes >= w.total, SGLK_ERR_WORKSPACE, "qkv_proj_with_rope: workspace %zu < required %zu",
                 a->workspace_bytes, w.total);
    unsigned char* ws = (unsigned char*)a->workspace;
    void* qa = ws + w.qa;
    void* qn = ws + w.qn;
    void* q2 = ws + w.q2;
    void* latent = ws + w.latent;
    const int qk = nope + rope;
    const bool i8 = a->wtype == SGLK_W_INT8;
    // one projection: x [B][K] bf16 -> out [B][N] bf16 (int8: per-token quantisation with the oracle's 1e-7 floor first)
    auto lin = [&](const void* x, int64_t x_stride, const void* wgt, const float* scale, int packed, void* out, int N, int K) -> int {
        sglk_scaled_mm_args m{};
        if (i8) {
            int rc = sglk_per_token_quant_int8_floor(x, x_stride, ws + w.xq, K, (float*)(ws + w.xs), B, K, 1e-7f, stream);
            if (rc != SGLK_OK) return rc;
            m.x = ws + w.xq;
            m.x_stride = K;
            m.x_is_int8 = 1;
            m.x_scale = (const float*)(ws + w.xs);
        } else {
            m.x = x;
            m.x_stride = x_stride;
        }
        m.w = wgt;
        m.w_scale = scale;
        m.out = out;
        m.out_stride = N;
        m.out_type = SGLK_OUT_BF16;
        m.M = B; m.N = N; m.K = K;
        m.wtype = a->wtype;
        m.packed = packed;
        m.block_n = a->wtype == SGLK_W_FP8_E4M3 ? a->block_n : 0;
        m.block_k = a->wtype == SGLK_W_FP8_E4M3 ? a->block_k : 0;
        m.workspace = ws + w.mm;
        m.workspace_bytes = w.mm_bytes;
        return sglk_scaled_mm(&m, stream);
    };
    int rc = lin(a->hidden, a->hidden_stride, a->q_a_w, a->q_a_scale, a->packed_q_a, qa, QL, hidden);
    if (rc != SGLK_OK) return rc;
    rc = sglk_rmsnorm(qn, QL, qa, QL, a->q_a_ln, B, QL, a->eps, 0, stream);
    if (rc != SGLK_OK) return rc;
    rc = lin(qn, QL, a->q_b_w, a->q_b_scale, a->packed_q_b, q2, H * qk, QL);
    if (rc != SGLK_OK) return rc;
    rc = lin(a->hidden, a->hidden_stride, a->kv_a_w, a->kv_a_scale, a->packed_kv_a, latent, R + rope, hidden);
    if (rc != SGLK_OK) return rc;
    rc = sglk_bmm_heads(q2, (int64_t)H * qk, qk, a->w_kc, a->w_kc_packed, a->q_input, a->q_stride_b, a->q_stride_h, B, H, R, nope, stream);
    if (rc != SGLK_OK) return rc;
    // v_input = rmsnorm(latent[:, :R]); k_input[:, :R] = the same rows, written by the same launch
    rc = launch_rmsnorm_bf16_dual(a->v_input, a->v_stride_b, a->k_input, a->k_stride_b, latent, R + rope, a->kv_a_ln, B, R, a->eps,
                                  (hipStream_t)stream);
    if (rc != SGLK_OK) return rc;
    const unsigned short* q_pe = (const unsigned short*)q2 + nope;
    const unsigned short* k_pe = (const unsigned short*)latent + R;
    return sglk_rope_gptj(q_pe, (int64_t)H * qk, qk, k_pe, R + rope, a->positions, a->positions_is64, a->cos_sin_cache, a->cache_stride,
                          (unsigned short*)a->q_input + R, a->q_stride_b, a->q_stride_h, (unsigned short*)a->k_input + R, a->k_stride_b,
                          B, H, rope, stream);
}
