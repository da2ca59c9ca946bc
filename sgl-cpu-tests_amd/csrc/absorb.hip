// Pieces of qkv_proj_with_rope (MLA "absorbed" projection, /root/reference/test_absorb.py:65-87,133-147) that are not
// a GEMM or an RMSNorm: the per-head product with w_kc and the GPT-J style rotary embedding.  The operator itself is
// composed in the Python layer (sgl_kernel/_ops.py) from these, sglk_scaled_mm and sglk_rmsnorm, mirroring the
// reference's op order and rounding points; oracle: oracle/absorb.py.
//
//   bmm_heads : out[b][h][oc] = sum_ic x[b][h][ic] * w[h][oc][ic]          (torch.bmm(q_nope^T, w_kc), test_absorb.py:73)
//               fp32 accumulation, one rounding to bf16.  w row-major [H][OC][IC] or in the packed (VNNI-2) order of
//               convert_weight_packed applied per head: [H][OC/32][IC/2][32][2].  Decode-sized: one thread per output,
//               weights streamed once (H * OC * IC * 2 B), coalesced in both layouts.
//   rope_gptj : q_out/k_out = x * cos + rotate(x) * sin, pairs (2i, 2i+1) share cache entries cos[i] = cache[pos][i],
//               sin[i] = cache[pos][d/2 + i]  (test_absorb.py:27-31,49-63); fp32 with separately rounded operations.
#include "sglk_common.h"

#pragma clang fp contract(off)

namespace sglk {
namespace {

template <bool PACKED>
__global__ __launch_bounds__(256) void bmm_heads_kernel(const uint16_t* __restrict__ x, int64_t x_sb, int64_t x_sh,
                                                        const uint16_t* __restrict__ w, uint16_t* __restrict__ out,
                                                        int64_t o_sb, int64_t o_sh, int B, int H, int OC, int IC) {
    extern __shared__ float xs[];                      // [kRows][IC] activations of this workgroup's rows, fp32
    constexpr int kRows = 4;
    const int h = blockIdx.x;
    const int oc = blockIdx.y * 64 + (threadIdx.x & 63);
    const int b0 = blockIdx.z * kRows, bl = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < kRows * IC; i += 256) {
        const int r = i / IC, c = i - r * IC;
        xs[i] = (b0 + r < B) ? bf16_bits_to_f32(x[(int64_t)(b0 + r) * x_sb + (int64_t)h * x_sh + c]) : 0.f;
    }
    __syncthreads();
    const int b = b0 + bl;
    if (oc >= OC || b >= B) return;
    const float* xr = xs + bl * IC;
    float acc = 0.f;
    if (PACKED) {
        // [H][OC/32][IC/2][32][2]: for a k pair p the 32 rows' dwords are consecutive
        const unsigned* wp = reinterpret_cast<const unsigned*>(w) + ((int64_t)h * (OC >> 5) + (oc >> 5)) * (IC >> 1) * 32 + (oc & 31);
        for (int p = 0; p < (IC >> 1); ++p) {
            const unsigned v = wp[(int64_t)p * 32];
            acc = __builtin_fmaf(xr[2 * p], __uint_as_float(v << 16), acc);
            acc = __builtin_fmaf(xr[2 * p + 1], __uint_as_float(v & 0xffff0000u), acc);
        }
    } else {
        const uint16_t* wr = w + ((int64_t)h * OC + oc) * IC;
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

// one thread per (row, pair): rows 0 .. B*H-1 are q rows (b, h), rows B*H .. B*H+B-1 the k rows
__global__ __launch_bounds__(256) void rope_gptj_kernel(const uint16_t* __restrict__ q_pe, int64_t q_sb, int64_t q_sh,
                                                        const uint16_t* __restrict__ k_pe, int64_t k_sb,
                                                        const void* __restrict__ pos, int pos_is64,
                                                        const uint16_t* __restrict__ cache, int64_t cache_stride,
                                                        uint16_t* __restrict__ q_out, int64_t qo_sb, int64_t qo_sh,
                                                        uint16_t* __restrict__ k_out, int64_t ko_sb, int B, int H, int D) {
    const int half = D >> 1;
    const int64_t total = ((int64_t)B * H + B) * half;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / half;
        const int pr = (int)(i - row * half);
        const uint16_t* src;
        uint16_t* dst;
        int b;
        if (row < (int64_t)B * H) {
            b = (int)(row / H);
            const int h = (int)(row - (int64_t)b * H);
            src = q_pe + (int64_t)b * q_sb + (int64_t)h * q_sh;
            dst = q_out + (int64_t)b * qo_sb + (int64_t)h * qo_sh;
        } else {
            b = (int)(row - (int64_t)B * H);
            src = k_pe + (int64_t)b * k_sb;
            dst = k_out + (int64_t)b * ko_sb;
        }
        const int64_t ps = pos_is64 ? reinterpret_cast<const int64_t*>(pos)[b] : (int64_t)reinterpret_cast<const int*>(pos)[b];
        const float c = bf16_bits_to_f32(cache[ps * cache_stride + pr]);
        const float s = bf16_bits_to_f32(cache[ps * cache_stride + half + pr]);
        const float x0 = bf16_bits_to_f32(src[2 * pr]), x1 = bf16_bits_to_f32(src[2 * pr + 1]);
        // x * cos + rotate(x) * sin with rotate(x) = (-x1, x0): two rounded products, one rounded sum each
        const float o0 = x0 * c + (-x1) * s;
        const float o1 = x1 * c + x0 * s;
        dst[2 * pr] = f32_to_bf16_bits(o0);
        dst[2 * pr + 1] = f32_to_bf16_bits(o1);
    }
}

}  // namespace

int launch_quant_int8_rows(const uint16_t* x, int64_t x_stride, int8_t* q, int64_t q_stride, float* scale, int64_t rows,
                           int cols, float floor, hipStream_t stream);

}  // namespace sglk

using namespace sglk;

extern "C" int sglk_bmm_heads(const void* x, int64_t x_stride_b, int64_t x_stride_h, const void* w, int32_t packed, void* out,
                              int64_t out_stride_b, int64_t out_stride_h, int32_t B, int32_t H, int32_t OC, int32_t IC,
                              void* stream) {
    SGLK_REQUIRE(B >= 0 && H > 0 && OC > 0 && IC > 0, SGLK_ERR_INVALID, "bmm_heads: bad sizes B=%d H=%d OC=%d IC=%d", B, H, OC, IC);
    if (B == 0) return SGLK_OK;
    SGLK_REQUIRE(x && w && out, SGLK_ERR_INVALID, "bmm_heads: null pointer");
    SGLK_REQUIRE(IC % 8 == 0 && IC <= 2048, SGLK_ERR_SHAPE, "bmm_heads: IC (%d) must be a multiple of 8 and <= 2048", IC);
    SGLK_REQUIRE(!packed || (OC % 32 == 0), SGLK_ERR_SHAPE, "bmm_heads: packed weights need OC (%d) %% 32 == 0", OC);
    SGLK_REQUIRE(packed || ((uintptr_t)w % 16) == 0, SGLK_ERR_INVALID, "bmm_heads: row-major weights must be 16-byte aligned");
    const dim3 grid((unsigned)H, (unsigned)ceil_div(OC, 64), (unsigned)ceil_div(B, 4)), block(256);
    const size_t lds = (size_t)4 * IC * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (packed)
        hipLaunchKernelGGL(bmm_heads_kernel<true>, grid, block, lds, s, (const uint16_t*)x, x_stride_b, x_stride_h,
                           (const uint16_t*)w, (uint16_t*)out, out_stride_b, out_stride_h, B, H, OC, IC);
    else
        hipLaunchKernelGGL(bmm_heads_kernel<false>, grid, block, lds, s, (const uint16_t*)x, x_stride_b, x_stride_h,
                           (const uint16_t*)w, (uint16_t*)out, out_stride_b, out_stride_h, B, H, OC, IC);
    SGLK_CHECK_LAUNCH("bmm_heads");
    return SGLK_OK;
}

extern "C" int sglk_rope_gptj(const void* q_pe, int64_t q_stride_b, int64_t q_stride_h, const void* k_pe, int64_t k_stride_b,
                              const void* positions, int32_t positions_is64, const void* cos_sin_cache, int64_t cache_stride,
                              void* q_out, int64_t q_out_stride_b, int64_t q_out_stride_h, void* k_out, int64_t k_out_stride_b,
                              int32_t B, int32_t H, int32_t rotary_dim, void* stream) {
    SGLK_REQUIRE(B >= 0 && H > 0 && rotary_dim > 0 && rotary_dim % 2 == 0, SGLK_ERR_INVALID,
                 "rope_gptj: bad sizes B=%d H=%d rotary_dim=%d", B, H, rotary_dim);
    if (B == 0) return SGLK_OK;
    SGLK_REQUIRE(q_pe && k_pe && positions && cos_sin_cache && q_out && k_out, SGLK_ERR_INVALID, "rope_gptj: null pointer");
    const int64_t total = ((int64_t)B * H + B) * (rotary_dim / 2);
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rope_gptj_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)q_pe,
                       q_stride_b, q_stride_h, (const uint16_t*)k_pe, k_stride_b, positions, positions_is64,
                       (const uint16_t*)cos_sin_cache, cache_stride, (uint16_t*)q_out, q_out_stride_b, q_out_stride_h,
                       (uint16_t*)k_out, k_out_stride_b, B, H, rotary_dim);
    SGLK_CHECK_LAUNCH("rope_gptj");
    return SGLK_OK;
}

extern "C" int sglk_per_token_quant_int8_floor(const void* x, int64_t x_stride, void* q, int64_t q_stride, float* scale,
                                               int64_t rows, int32_t cols, float floor, void* stream) {
    SGLK_REQUIRE(rows >= 0 && cols > 0 && floor > 0.f, SGLK_ERR_INVALID, "per_token_quant_int8: bad sizes");
    if (rows == 0) return SGLK_OK;
    SGLK_REQUIRE(x && q && scale, SGLK_ERR_INVALID, "per_token_quant_int8: null pointer");
    return launch_quant_int8_rows((const uint16_t*)x, x_stride, (int8_t*)q, q_stride, scale, rows, cols, floor, (hipStream_t)stream);
}
