// HBM-bound row kernels: silu_and_mul, rmsnorm, fused_add_rmsnorm (bf16 / fp16).
//
// Oracles: /root/reference/test_activation.py:14-16 (F.silu(x[:d]) * x[d:]),
//          /root/reference/test_norm.py:15-33 (fp32 variance, x*rsqrt rounded to the I/O dtype, THEN times weight
//          in the I/O dtype; fused-add variant writes the fp32 sum back to `residual` rounded once).
// One workgroup per row, 16-byte accesses whenever the row is 16-byte aligned and a multiple of 8 wide, scalar
// fallback otherwise (hidden size 4109 is one of the reference's shapes).  fp32 math, roundings placed where the
// reference's torch code places them.
#include "knobs.h"

namespace sglk {

template <bool F16>
SGLK_DEV float ld_elem(const unsigned short* p) {
    if (F16) return (float)__builtin_bit_cast(_Float16, *p);
    return bf16_bits_to_f32(*p);
}
template <bool F16>
SGLK_DEV unsigned short to_bits(float v) {
    if (F16) return __builtin_bit_cast(unsigned short, (_Float16)v);
    return f32_to_bf16_bits(v);
}
template <bool F16>
SGLK_DEV float round_io(float v) {   // value after one rounding to the I/O dtype
    if (F16) return (float)(_Float16)v;
    return bf16_bits_to_f32(f32_to_bf16_bits(v));
}
template <bool F16>
SGLK_DEV void unpack8(const uint4& v, float (&f)[8]) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (F16) {
            f[2 * j] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[j] & 0xffff));
            f[2 * j + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[j] >> 16));
        } else {
            f[2 * j] = __uint_as_float(w[j] << 16);
            f[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
        }
    }
}
template <bool F16>
SGLK_DEV uint4 pack8(const float (&f)[8]) {
    unsigned w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = (unsigned)to_bits<F16>(f[2 * j]) | ((unsigned)to_bits<F16>(f[2 * j + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- silu_and_mul ------------------------------------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256) void silu_and_mul_kernel(const unsigned short* __restrict__ x, int64_t x_stride,
                                                           unsigned short* __restrict__ out, int64_t out_stride,
                                                           int64_t rows, int d) {
    const bool vec = (d % 8 == 0) && (x_stride % 8 == 0) && (out_stride % 8 == 0) &&
                     ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const unsigned short* xr = x + row * x_stride;
        unsigned short* orow = out + row * out_stride;
        if (vec) {
            for (int c = threadIdx.x * 8; c < d; c += 256 * 8) {
                float g[8], u[8], o[8];
                unpack8<F16>(*reinterpret_cast<const uint4*>(xr + c), g);
                unpack8<F16>(*reinterpret_cast<const uint4*>(xr + d + c), u);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = round_io<F16>(silu_f32(g[j])) * u[j];   // torch rounds silu() before the multiply
                *reinterpret_cast<uint4*>(orow + c) = pack8<F16>(o);
            }
        } else {
            for (int c = threadIdx.x; c < d; c += 256)
                orow[c] = to_bits<F16>(round_io<F16>(silu_f32(ld_elem<F16>(xr + c))) * ld_elem<F16>(xr + d + c));
        }
    }
}

// ---- rmsnorm / fused_add_rmsnorm -------------------------------------------------------------------------------------
SGLK_DEV float block_sum_256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6;
    __syncthreads();   // red may still be read from the previous row
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// FUSED: x <- rmsnorm(x + residual) * w, residual <- x + residual (both in place); else out <- rmsnorm(x) * w
template <bool F16, bool FUSED>
__global__ __launch_bounds__(256) void rmsnorm_kernel(unsigned short* __restrict__ out, int64_t out_stride,
                                                      const unsigned short* __restrict__ x, int64_t x_stride,
                                                      unsigned short* __restrict__ residual, int64_t res_stride,
                                                      const unsigned short* __restrict__ w, int64_t rows, int h,
                                                      float eps, unsigned short* __restrict__ out2 = nullptr,
                                                      int64_t out2_stride = 0) {   // out2: a second copy of the result rows
    __shared__ float red[4];
    extern __shared__ __attribute__((aligned(16))) float rowbuf[];   // FUSED: the fp32 sum of the row
    const bool vec = (h % 8 == 0) && (x_stride % 8 == 0) && (out_stride % 8 == 0) && (!FUSED || res_stride % 8 == 0) &&
                     ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(w) & 15) == 0) &&
                     (!FUSED || (reinterpret_cast<uintptr_t>(residual) & 15) == 0) &&
                     (!out2 || (out2_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(out2) & 15) == 0));
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const unsigned short* xr = x + row * x_stride;
        unsigned short* orow = out + row * out_stride;
        unsigned short* rr = FUSED ? residual + row * res_stride : nullptr;
        float ss = 0.f;
        if (vec) {
            for (int c = threadIdx.x * 8; c < h; c += 256 * 8) {
                float f[8];
                unpack8<F16>(*reinterpret_cast<const uint4*>(xr + c), f);
                if (FUSED) {
                    float r[8];
                    unpack8<F16>(*reinterpret_cast<const uint4*>(rr + c), r);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { f[j] += r[j]; rowbuf[c + j] = f[j]; }
                    *reinterpret_cast<uint4*>(rr + c) = pack8<F16>(f);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
            }
        } else {
            for (int c = threadIdx.x; c < h; c += 256) {
                float f = ld_elem<F16>(xr + c);
                if (FUSED) {
                    f += ld_elem<F16>(rr + c);
                    rowbuf[c] = f;
                    rr[c] = to_bits<F16>(f);
                }
                ss += f * f;
            }
        }
        const float var = block_sum_256(ss, red) / (float)h;
        const float inv = rsqrtf(var + eps);
        if (vec) {
            for (int c = threadIdx.x * 8; c < h; c += 256 * 8) {
                float f[8], wv[8], o[8];
                if (FUSED) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = rowbuf[c + j];
                } else {
                    unpack8<F16>(*reinterpret_cast<const uint4*>(xr + c), f);
                }
                unpack8<F16>(*reinterpret_cast<const uint4*>(w + c), wv);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = round_io<F16>(f[j] * inv) * wv[j];
                const uint4 packed = pack8<F16>(o);
                *reinterpret_cast<uint4*>(orow + c) = packed;
                if (out2) *reinterpret_cast<uint4*>(out2 + row * out2_stride + c) = packed;
            }
        } else {
            for (int c = threadIdx.x; c < h; c += 256) {
                const float f = FUSED ? rowbuf[c] : ld_elem<F16>(xr + c);
                const unsigned short bits = to_bits<F16>(round_io<F16>(f * inv) * ld_elem<F16>(w + c));
                orow[c] = bits;
                if (out2) out2[row * out2_stride + c] = bits;
            }
        }
    }
}

}  // namespace sglk

using namespace sglk;

static int64_t row_grid(int64_t rows) { return rows < 256 * 8 ? rows : 256 * 8; }

extern "C" int sglk_silu_and_mul(const void* x, int64_t x_stride, void* out, int64_t out_stride, int64_t rows, int32_t d,
                                 int32_t is_f16, void* stream) {
    SGLK_REQUIRE(rows >= 0 && d > 0, SGLK_ERR_INVALID, "silu_and_mul: bad sizes rows=%lld d=%d", (long long)rows, d);
    SGLK_REQUIRE(rows == 0 || (x && out), SGLK_ERR_INVALID, "silu_and_mul: null pointer");
    SGLK_REQUIRE(x_stride >= 2 * (int64_t)d && out_stride >= d, SGLK_ERR_INVALID, "silu_and_mul: stride too small");
    if (rows == 0) return SGLK_OK;
    const dim3 grid((unsigned)row_grid(rows)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (is_f16)
        hipLaunchKernelGGL(silu_and_mul_kernel<true>, grid, block, 0, s, (const unsigned short*)x, x_stride,
                           (unsigned short*)out, out_stride, rows, d);
    else
        hipLaunchKernelGGL(silu_and_mul_kernel<false>, grid, block, 0, s, (const unsigned short*)x, x_stride,
                           (unsigned short*)out, out_stride, rows, d);
    SGLK_CHECK_LAUNCH("silu_and_mul");
    return SGLK_OK;
}

extern "C" int sglk_rmsnorm(void* out, int64_t out_stride, const void* x, int64_t x_stride, const void* weight,
                            int64_t rows, int32_t hidden, float eps, int32_t is_f16, void* stream) {
    SGLK_REQUIRE(rows >= 0 && hidden > 0, SGLK_ERR_INVALID, "rmsnorm: bad sizes");
    SGLK_REQUIRE(rows == 0 || (x && out && weight), SGLK_ERR_INVALID, "rmsnorm: null pointer");
    SGLK_REQUIRE(x_stride >= hidden && out_stride >= hidden, SGLK_ERR_INVALID, "rmsnorm: stride < hidden");
    if (rows == 0) return SGLK_OK;
    const dim3 grid((unsigned)row_grid(rows)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (is_f16)
        hipLaunchKernelGGL((rmsnorm_kernel<true, false>), grid, block, 0, s, (unsigned short*)out, out_stride,
                           (const unsigned short*)x, x_stride, (unsigned short*)nullptr, (int64_t)0,
                           (const unsigned short*)weight, rows, hidden, eps);
    else
        hipLaunchKernelGGL((rmsnorm_kernel<false, false>), grid, block, 0, s, (unsigned short*)out, out_stride,
                           (const unsigned short*)x, x_stride, (unsigned short*)nullptr, (int64_t)0,
                           (const unsigned short*)weight, rows, hidden, eps);
    SGLK_CHECK_LAUNCH("rmsnorm");
    return SGLK_OK;
}

// rmsnorm of bf16 rows written to two places (qkv_proj_with_rope: v_input and the head of k_input are the same rows)
namespace sglk {
int launch_rmsnorm_bf16_dual(void* out, int64_t out_stride, void* out2, int64_t out2_stride, const void* x, int64_t x_stride,
                             const void* weight, int64_t rows, int hidden, float eps, hipStream_t stream) {
    if (rows == 0) return SGLK_OK;
    hipLaunchKernelGGL((rmsnorm_kernel<false, false>), dim3((unsigned)row_grid(rows)), dim3(256), 0, stream, (unsigned short*)out,
                       out_stride, (const unsigned short*)x, x_stride, (unsigned short*)nullptr, (int64_t)0,
                       (const unsigned short*)weight, rows, hidden, eps, (unsigned short*)out2, out2_stride);
    SGLK_CHECK_LAUNCH("rmsnorm(dual)");
    return SGLK_OK;
}
}  // namespace sglk

extern "C" int sglk_fused_add_rmsnorm(void* x, int64_t x_stride, void* residual, int64_t res_stride, const void* weight,
                                      int64_t rows, int32_t hidden, float eps, int32_t is_f16, void* stream) {
    SGLK_REQUIRE(rows >= 0 && hidden > 0, SGLK_ERR_INVALID, "fused_add_rmsnorm: bad sizes");
    SGLK_REQUIRE(rows == 0 || (x && residual && weight), SGLK_ERR_INVALID, "fused_add_rmsnorm: null pointer");
    SGLK_REQUIRE(x_stride >= hidden && res_stride >= hidden, SGLK_ERR_INVALID, "fused_add_rmsnorm: stride < hidden");
    SGLK_REQUIRE((size_t)hidden * 4 <= 150 * 1024, SGLK_ERR_SHAPE, "fused_add_rmsnorm: hidden %d too wide for one LDS row", hidden);
    if (rows == 0) return SGLK_OK;
    const dim3 grid((unsigned)row_grid(rows)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)hidden * 4;
    if (is_f16) {
        if (lds > 48 * 1024) SGLK_ENSURE_DYN_LDS((rmsnorm_kernel<true, true>), 150 * 1024, "fused_add_rmsnorm");
        hipLaunchKernelGGL((rmsnorm_kernel<true, true>), grid, block, lds, s, (unsigned short*)x, x_stride,
                           (const unsigned short*)x, x_stride, (unsigned short*)residual, res_stride,
                           (const unsigned short*)weight, rows, hidden, eps);
    } else {
        if (lds > 48 * 1024) SGLK_ENSURE_DYN_LDS((rmsnorm_kernel<false, true>), 150 * 1024, "fused_add_rmsnorm");
        hipLaunchKernelGGL((rmsnorm_kernel<false, true>), grid, block, lds, s, (unsigned short*)x, x_stride,
                           (const unsigned short*)x, x_stride, (unsigned short*)residual, res_stride,
                           (const unsigned short*)weight, rows, hidden, eps);
    }
    SGLK_CHECK_LAUNCH("fused_add_rmsnorm");
    return SGLK_OK;
}
