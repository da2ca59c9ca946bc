// qkv_proj_with_rope (/root/reference/test_absorb.py:133-147,184-186; oracles native_torch :65-87, native_torch_int8 :89-131)
// as ONE C-ABI call: the MLA "absorbed" projection
//     q = rmsnorm(hidden . q_a^T) . q_b^T        latent = hidden . kv_a^T
//     q_input[b, h] = [ q_nope[b, h] . w_kc[h] | rope(q_pe[b, h]) ]
//     v_input[b] = rmsnorm(latent[b, :R]);  k_input[b] = [ v_input[b] | rope(k_pe[b]) ]
// The steps are this library's own entry points (sglk_scaled_mm for the three projections in bf16 / fp8 W8A16 / int8 W8A8,
// sglk_rmsnorm, sglk_bmm_heads, sglk_rope_gptj, sglk_per_token_quant_int8_floor), in the oracle's order with the oracle's rounding
// points (every intermediate is a bf16 tensor, as in native_torch) -- the call exists so that a decode step costs one host call and
// one workspace instead of eleven calls and a dozen allocations; intermediates live in the caller's workspace.
//
// Five launches at decode sizes: the three projections keep their split-K partials (set_splitk_capture), and the work between
// them is two kernels that start by summing those partials --
//   qkv_mid_kernel : q_a partials -> bf16 -> RMSNorm -> qn;   kv_a partials -> bf16 latent -> RMSNorm -> v_input and the head
//                    of k_input, rotary embedding of k_pe -> the tail of k_input          (was: 2 reduces, 2 norms, a copy, rope)
//   qkv_tail_kernel: q_b partials -> bf16 q; q_nope . w_kc -> q_input[..., :R]; rotary embedding of q_pe -> q_input[..., R:]
//                                                                                         (was: a reduce, bmm_heads, rope)
// A projection whose kernel does not split (or reduces in its own way: int8) hands over its bf16 output instead; the two
// kernels take either.
#include "moe_internal.h"
#include "sglk.h"

#pragma clang fp contract(off)   // rotary embedding: separately rounded products and sums, like absorb.hip's

namespace sglk {
namespace {

struct RowSrc {                 // a [rows][n] bf16 matrix, or the fp32 split-K partials it is the rounded sum of
    const float* partial;       // [ks][rows][n] or nullptr
    int ks;
    int64_t rows;
    int n;
    const unsigned short* dense;
    int64_t dense_stride;
};
// Two elements (row, column) at a time (c1 < 0: one).  Ascending ranges, as the reduce kernel sums them.  All the loads of a batch of BATCH
// ranges -- of BOTH columns -- are in flight together: a range past the end re-reads the last one and is not added.  (Eight per batch,
// a serial tail, and one column after the other made these two kernels a chain of five to ten memory round trips: 7.4 and 5.9 us at
// one token, more than any of the three projections between them.)
template <int BATCH>
SGLK_DEV void row_val2_b(const RowSrc& s, int64_t r0, int c0, int64_t r1, int c1, float& v0, float& v1) {
    const bool two = c1 >= 0;
    if (s.partial) {
        const float* p0 = s.partial + r0 * s.n + c0;
        const float* p1 = two ? s.partial + r1 * s.n + c1 : p0;
        const int64_t step = s.rows * s.n;
        float a0 = 0.f, a1 = 0.f;
        for (int k = 0; k < s.ks; k += BATCH) {
            float t0[BATCH], t1[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int kk = k + j < s.ks ? k + j : s.ks - 1;
                t0[j] = p0[(int64_t)kk * step];
                t1[j] = p1[(int64_t)kk * step];
            }
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                if (k + j < s.ks) {
                    a0 += t0[j];
                    a1 += t1[j];
                }
            }
        }
        v0 = bf16_bits_to_f32(f32_to_bf16_bits(a0));                                          // the projection's bf16 output
        v1 = bf16_bits_to_f32(f32_to_bf16_bits(a1));
        return;
    }
    v0 = bf16_bits_to_f32(s.dense[r0 * s.dense_stride + c0]);
    v1 = two ? bf16_bits_to_f32(s.dense[r1 * s.dense_stride + c1]) : 0.f;
}
SGLK_DEV void row_val2(const RowSrc& s, int64_t r0, int c0, int64_t r1, int c1, float& v0, float& v1) {
    if (s.ks <= 8) row_val2_b<8>(s, r0, c0, r1, c1, v0, v1);      // few ranges (q_b): no sixteen-wide batch of mostly repeated loads
    else row_val2_b<16>(s, r0, c0, r1, c1, v0, v1);
}
SGLK_DEV float block_sum_1024(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i];
    return t;
}
SGLK_DEV void rope_pair(float x0, float x1, const unsigned short* cache_row, int pr, int half, unsigned short* dst) {
    const float c = bf16_bits_to_f32(cache_row[pr]), s = bf16_bits_to_f32(cache_row[half + pr]);
    const float o0 = x0 * c + (-x1) * s;        // x * cos + rotate(x) * sin, rotate(x) = (-x1, x0)
    const float o1 = x1 * c + x0 * s;
    dst[2 * pr] = f32_to_bf16_bits(o0);
    dst[2 * pr + 1] = f32_to_bf16_bits(o1);
}

// workgroups [0, B): row b of the q path; [B, 2B): row b of the kv path.  rmsnorm as elementwise.hip's: bf16(x * inv) * w -> bf16
__global__ __launch_bounds__(1024) void qkv_mid_kernel(RowSrc qa, RowSrc lat, const unsigned short* __restrict__ ln1,
                                                      const unsigned short* __restrict__ ln2, float eps,
                                                      unsigned short* __restrict__ qn, int64_t qn_stride,
                                                      unsigned short* __restrict__ v_out, int64_t v_stride,
                                                      unsigned short* __restrict__ k_out, int64_t k_stride, const void* pos,
                                                      int pos_is64, const unsigned short* __restrict__ cache, int64_t cache_stride,
                                                      int B, int QL, int R, int rope) {
    extern __shared__ float rowbuf[];
    __shared__ float red[16];
    const bool kv = (int)blockIdx.x >= B;
    const int b = kv ? blockIdx.x - B : blockIdx.x;
    const RowSrc& src = kv ? lat : qa;
    const int width = kv ? R + rope : QL, normed = kv ? R : QL;
    float ss = 0.f;
    for (int c = threadIdx.x; c < width; c += 2048) {
        const int c1 = c + 1024 < width ? c + 1024 : -1;
        float f0, f1;
        row_val2(src, b, c, b, c1, f0, f1);
        rowbuf[c] = f0;
        if (c < normed) ss += f0 * f0;
        if (c1 >= 0) {
            rowbuf[c1] = f1;
            if (c1 < normed) ss += f1 * f1;
        }
    }
    const float var = block_sum_1024(ss, red) / (float)normed;     // the barriers inside also publish rowbuf
    const float inv = rsqrtf(var + eps);
    const unsigned short* w = kv ? ln2 : ln1;
    for (int c = threadIdx.x; c < normed; c += 1024) {
        const float o = bf16_bits_to_f32(f32_to_bf16_bits(rowbuf[c] * inv)) * bf16_bits_to_f32(w[c]);
        const unsigned short bits = f32_to_bf16_bits(o);
        if (kv) {
            v_out[(int64_t)b * v_stride + c] = bits;
            k_out[(int64_t)b * k_stride + c] = bits;
        } else {
            qn[(int64_t)b * qn_stride + c] = bits;
        }
    }
    if (kv && (int)threadIdx.x < rope / 2) {
        const int64_t ps = pos_is64 ? reinterpret_cast<const int64_t*>(pos)[b] : (int64_t)reinterpret_cast<const int*>(pos)[b];
        const int pr = threadIdx.x;
        rope_pair(rowbuf[R + 2 * pr], rowbuf[R + 2 * pr + 1], cache + ps * cache_stride, pr, rope / 2, k_out + (int64_t)b * k_stride + R);
    }
}

// grid (H, ceil(R / 64), ceil(B / 4)): q_input[b][h][oc] = sum_ic q_nope[b][h][ic] * w_kc[h][oc][ic] (fp32 fma chain in ic order,
// one rounding: absorb.hip's bmm_heads) and, in the oc-block-0 workgroups, q_input[b][h][R:] = rope(q_pe[b][h])
template <bool PACKED>
__global__ __launch_bounds__(256) void qkv_tail_kernel(RowSrc q2, const unsigned short* __restrict__ w,
                                                       unsigned short* __restrict__ out, int64_t o_sb, int64_t o_sh, const void* pos,
                                                       int pos_is64, const unsigned short* __restrict__ cache, int64_t cache_stride,
                                                       int B, int H, int OC, int IC, int rope) {
    extern __shared__ float xs[];                      // [4][IC + rope] q values of this workgroup's rows, fp32
    constexpr int kRows = 4;
    const int qk = IC + rope;
    const int h = blockIdx.x;
    const int oc = blockIdx.y * 64 + (threadIdx.x & 63);
    const int b0 = blockIdx.z * kRows, bl = threadIdx.x >> 6;
    const int ncol = blockIdx.y == 0 ? qk : IC;        // only the rope workgroups need the pe part
    for (int i = threadIdx.x; i < kRows * ncol; i += 512) {
        const int i1 = i + 256 < kRows * ncol ? i + 256 : -1;
        const int r0 = i / ncol, c0 = i - r0 * ncol;
        const int r1 = i1 >= 0 ? i1 / ncol : r0, c1 = i1 >= 0 ? i1 - r1 * ncol : c0;
        // rows past the batch read row B - 1 (valid memory) and store zeros
        const int rr0 = b0 + r0 < B ? b0 + r0 : B - 1, rr1 = b0 + r1 < B ? b0 + r1 : B - 1;
        float f0, f1;
        row_val2(q2, rr0, h * qk + c0, rr1, i1 >= 0 ? h * qk + c1 : -1, f0, f1);
        xs[r0 * qk + c0] = b0 + r0 < B ? f0 : 0.f;
        if (i1 >= 0) xs[r1 * qk + c1] = b0 + r1 < B ? f1 : 0.f;
    }
    __syncthreads();
    if (blockIdx.y == 0 && (int)threadIdx.x < kRows * (rope / 2)) {
        const int r = threadIdx.x / (rope / 2), pr = threadIdx.x - r * (rope / 2);
        const int b = b0 + r;
        if (b < B) {
            const int64_t ps = pos_is64 ? reinterpret_cast<const int64_t*>(pos)[b] : (int64_t)reinterpret_cast<const int*>(pos)[b];
            rope_pair(xs[r * qk + IC + 2 * pr], xs[r * qk + IC + 2 * pr + 1], cache + ps * cache_stride, pr, rope / 2,
                      out + (int64_t)b * o_sb + (int64_t)h * o_sh + OC);
        }
    }
    const int b = b0 + bl;
    if (oc >= OC || b >= B) return;
    const float* xr = xs + bl * qk;
    float acc = 0.f;
    if (PACKED) {
        const unsigned* wp = reinterpret_cast<const unsigned*>(w) + ((int64_t)h * (OC >> 5) + (oc >> 5)) * (IC >> 1) * 32 + (oc & 31);
        for (int p = 0; p < (IC >> 1); ++p) {
            const unsigned v = wp[(int64_t)p * 32];
            acc = __builtin_fmaf(xr[2 * p], __uint_as_float(v << 16), acc);
            acc = __builtin_fmaf(xr[2 * p + 1], __uint_as_float(v & 0xffff0000u), acc);
        }
    } else {
        const unsigned short* wr = w + ((int64_t)h * OC + oc) * IC;
        for (int c = 0; c < IC; c += 8) {
            const uint4 v = *reinterpret_cast<const uint4*>(wr + c);
            const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc = __builtin_fmaf(xr[c + 2 * q], __uint_as_float(u[q] << 16), acc);
                acc = __builtin_fmaf(xr[c + 2 * q + 1], __uint_as_float(u[q] & 0xffff0000u), acc);
            }
        }
    }
    out[(int64_t)b * o_sb + (int64_t)h * o_sh + oc] = f32_to_bf16_bits(acc);
}

struct QkvWs {
    size_t qa, qn, q2, latent, xq, xs, mm, mm2, total, mm_bytes;
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
    w.mm = take(m);      // q_a, then q_b (its partials are consumed before q_b runs)
    w.mm2 = take(m);     // kv_a: its partials live beside q_a's until qkv_mid_kernel has run
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
    auto lin = [&](const void* x, int64_t x_stride, const void* wgt, const float* scale, int packed, void* out, int N, int K,
                   size_t mm_off, SplitkCapture* cap) -> int {
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
        m.workspace = ws + mm_off;
        m.workspace_bytes = w.mm_bytes;
        cap->partial = nullptr;
        set_splitk_capture(cap);                 // keep the split-K partials: the next kernel sums them itself
        const int rc2 = sglk_scaled_mm(&m, stream);
        set_splitk_capture(nullptr);
        return rc2;
    };
    auto src_of = [&](const SplitkCapture& c, const void* dense, int n) {
        RowSrc r{};
        r.partial = c.partial;
        r.ks = c.ksplit;
        r.rows = c.rows;
        r.n = n;
        r.dense = (const unsigned short*)dense;
        r.dense_stride = n;
        return r;
    };
    hipStream_t s = (hipStream_t)stream;
    SplitkCapture c1{}, c2{}, c3{};
    // (kv_a on a second stream beside q_a -- fork and join inside the call -- was built and measured: 30.5-31.1 us either way at one
    // token in hipGraph replays, so the call stays on one stream.)
    int rc = lin(a->hidden, a->hidden_stride, a->q_a_w, a->q_a_scale, a->packed_q_a, qa, QL, hidden, w.mm, &c1);
    if (rc != SGLK_OK) return rc;
    rc = lin(a->hidden, a->hidden_stride, a->kv_a_w, a->kv_a_scale, a->packed_kv_a, latent, R + rope, hidden, w.mm2, &c2);
    if (rc != SGLK_OK) return rc;
    SGLK_REQUIRE((!c1.partial || (c1.rows == B && c1.n == QL)) && (!c2.partial || (c2.rows == B && c2.n == R + rope)), SGLK_ERR_INVALID,
                 "qkv_proj_with_rope: unexpected split-K layout");
    const int wmax = QL > R + rope ? QL : R + rope;
    hipLaunchKernelGGL(qkv_mid_kernel, dim3((unsigned)(2 * B)), dim3(1024), (size_t)wmax * sizeof(float), s, src_of(c1, qa, QL),
                       src_of(c2, latent, R + rope), (const unsigned short*)a->q_a_ln, (const unsigned short*)a->kv_a_ln, a->eps,
                       (unsigned short*)qn, (int64_t)QL, (unsigned short*)a->v_input, a->v_stride_b, (unsigned short*)a->k_input,
                       a->k_stride_b, a->positions, a->positions_is64, (const unsigned short*)a->cos_sin_cache, a->cache_stride, B, QL,
                       R, rope);
    SGLK_CHECK_LAUNCH("qkv_proj_with_rope(mid)");
    rc = lin(qn, QL, a->q_b_w, a->q_b_scale, a->packed_q_b, q2, H * qk, QL, w.mm, &c3);
    if (rc != SGLK_OK) return rc;
    SGLK_REQUIRE(!c3.partial || (c3.rows == B && c3.n == H * qk), SGLK_ERR_INVALID, "qkv_proj_with_rope: unexpected split-K layout");
    SGLK_REQUIRE(nope % 8 == 0 && nope <= 2048 && (!a->w_kc_packed || R % 32 == 0), SGLK_ERR_SHAPE,
                 "qkv_proj_with_rope: nope (%d) must be a multiple of 8, packed w_kc needs kv_lora (%d) %% 32 == 0", nope, R);
    SGLK_REQUIRE(a->w_kc_packed || ((uintptr_t)a->w_kc % 16) == 0, SGLK_ERR_INVALID, "qkv_proj_with_rope: w_kc must be 16-byte aligned");
    SGLK_REQUIRE(rope / 2 * 4 <= 256, SGLK_ERR_SHAPE, "qkv_proj_with_rope: rope dim %d too large", rope);
    const dim3 grid((unsigned)H, (unsigned)ceil_div(R, 64), (unsigned)ceil_div(B, 4));
    const size_t lds = (size_t)4 * qk * sizeof(float);
    if (a->w_kc_packed)
        hipLaunchKernelGGL(qkv_tail_kernel<true>, grid, dim3(256), lds, s, src_of(c3, q2, H * qk), (const unsigned short*)a->w_kc,
                           (unsigned short*)a->q_input, a->q_stride_b, a->q_stride_h, a->positions, a->positions_is64,
                           (const unsigned short*)a->cos_sin_cache, a->cache_stride, B, H, R, nope, rope);
    else
        hipLaunchKernelGGL(qkv_tail_kernel<false>, grid, dim3(256), lds, s, src_of(c3, q2, H * qk), (const unsigned short*)a->w_kc,
                           (unsigned short*)a->q_input, a->q_stride_b, a->q_stride_h, a->positions, a->positions_is64,
                           (const unsigned short*)a->cos_sin_cache, a->cache_stride, B, H, R, nope, rope);
    SGLK_CHECK_LAUNCH("qkv_proj_with_rope(tail)");
    return SGLK_OK;
}
