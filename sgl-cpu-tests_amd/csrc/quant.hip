// Per-row symmetric int8 quantisation of bf16 activations.
// Oracle: /root/reference/test_gemm_int8.py:14-22 (floor 1e-10), /root/reference/test_moe_int8.py:23-31 (floor 1e-7):
//   amax = max(|row|, floor); scale = amax / 127; q = round_half_even(x * (127 / amax))
// One wave per row, 16-byte loads, wave reduction; HBM-bound.
#include "moe_internal.h"

namespace sglk {

__global__ __launch_bounds__(256) void quant_int8_rows_kernel(const uint16_t* __restrict__ x, int64_t x_stride,
                                                              int8_t* __restrict__ q, int64_t q_stride,
                                                              float* __restrict__ scale, int64_t rows, int cols,
                                                              float floor_v) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint16_t* xr = x + row * x_stride;
    int8_t* qr = q + row * q_stride;
    const bool vec = (cols % 8 == 0) && ((reinterpret_cast<uintptr_t>(xr) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(qr) & 7) == 0);
    float amax = 0.f;
    if (vec) {
        for (int c = lane * 8; c < cols; c += 64 * 8) {
            const uint4 v = *reinterpret_cast<const uint4*>(xr + c);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                amax = fmaxf(amax, fabsf(__uint_as_float(w[j] << 16)));
                amax = fmaxf(amax, fabsf(__uint_as_float(w[j] & 0xffff0000u)));
            }
        }
    } else {
        for (int c = lane; c < cols; c += 64) amax = fmaxf(amax, fabsf(bf16_bits_to_f32(xr[c])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    amax = fmaxf(amax, floor_v);
    // torch evaluates `127 / absmax` as reciprocal(absmax) * 127 (Tensor.__rdiv__): two roundings, reproduced here
    const float inv = (1.0f / amax) * 127.0f;
    if (lane == 0) scale[row] = amax / 127.0f;
    if (vec) {
        for (int c = lane * 8; c < cols; c += 64 * 8) {
            const uint4 v = *reinterpret_cast<const uint4*>(xr + c);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
            unsigned out[2] = {0u, 0u};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int a = (int)rintf(__uint_as_float(w[j] << 16) * inv);
                const int b = (int)rintf(__uint_as_float(w[j] & 0xffff0000u) * inv);
                out[j >> 1] |= ((unsigned)(a & 0xff) << (16 * (j & 1))) | ((unsigned)(b & 0xff) << (16 * (j & 1) + 8));
            }
            *reinterpret_cast<uint2*>(qr + c) = make_uint2(out[0], out[1]);
        }
    } else {
        for (int c = lane; c < cols; c += 64) qr[c] = (int8_t)(int)rintf(bf16_bits_to_f32(xr[c]) * inv);
    }
}

// fp32 rows (the W8A8 MoE quantises SiLU*mul output straight from fp32, like the oracle test_moe_int8.py:83-86)
__global__ __launch_bounds__(256) void quant_int8_rows_f32_kernel(const float* __restrict__ x, int64_t x_stride,
                                                                  int8_t* __restrict__ q, int64_t q_stride,
                                                                  float* __restrict__ scale, int64_t rows, int cols,
                                                                  float floor_v) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * x_stride;
    int8_t* qr = q + row * q_stride;
    float amax = 0.f;
    for (int c = lane; c < cols; c += 64) amax = fmaxf(amax, fabsf(xr[c]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    amax = fmaxf(amax, floor_v);
    // torch evaluates `127 / absmax` as reciprocal(absmax) * 127 (Tensor.__rdiv__): two roundings, reproduced here
    const float inv = (1.0f / amax) * 127.0f;
    if (lane == 0) scale[row] = amax / 127.0f;
    for (int c = lane; c < cols; c += 64) qr[c] = (int8_t)(int)rintf(xr[c] * inv);
}

int launch_quant_int8_rows_f32(const float* x, int64_t x_stride, int8_t* q, int64_t q_stride, float* scale,
                               int64_t rows, int cols, float floor_v, hipStream_t stream) {
    if (rows == 0) return SGLK_OK;
    hipLaunchKernelGGL(quant_int8_rows_f32_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, stream, x, x_stride,
                       q, q_stride, scale, rows, cols, floor_v);
    SGLK_CHECK_LAUNCH("quant_int8_rows_f32");
    return SGLK_OK;
}

int launch_quant_int8_rows(const uint16_t* x, int64_t x_stride, int8_t* q, int64_t q_stride, float* scale, int64_t rows,
                           int cols, float floor_v, hipStream_t stream) {
    if (rows == 0) return SGLK_OK;
    const int64_t blocks = ceil_div(rows, 4);
    hipLaunchKernelGGL(quant_int8_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, x_stride, q, q_stride,
                       scale, rows, cols, floor_v);
    SGLK_CHECK_LAUNCH("quant_int8_rows");
    return SGLK_OK;
}

}  // namespace sglk

extern "C" int sglk_per_token_quant_int8(const void* x, int64_t x_stride, void* q, int64_t q_stride, float* scale,
                                         int64_t rows, int32_t cols, void* stream) {
    using namespace sglk;
    SGLK_REQUIRE(rows >= 0 && cols > 0, SGLK_ERR_INVALID, "per_token_quant_int8: bad sizes");
    SGLK_REQUIRE(rows == 0 || (x && q && scale), SGLK_ERR_INVALID, "per_token_quant_int8: null pointer");
    SGLK_REQUIRE(x_stride >= cols && q_stride >= cols, SGLK_ERR_INVALID, "per_token_quant_int8: stride < cols");
    return launch_quant_int8_rows((const uint16_t*)x, x_stride, (int8_t*)q, q_stride, scale, rows, cols, 1e-10f,
                                  (hipStream_t)stream);
}
