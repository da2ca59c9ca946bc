// Per-row symmetric int8 quantisation of bf16 activations.
// Oracle: /root/reference/test_gemm_int8.py:14-22 (floor 1e-10), /root/reference/test_moe_int8.py:23-31 (floor 1e-7):
//   amax = max(|row|, floor); scale = amax / 127; q = round_half_even(x * (127 / amax))
// One wave per row, 16-byte loads, wave reduction; HBM-bound.
#include "moe_internal.h"
#include "quant_rows.h"

namespace sglk {

__global__ __launch_bounds__(256) void quant_int8_rows_kernel(const uint16_t* __restrict__ x, int64_t x_stride,
                                                              int8_t* __restrict__ q, int64_t q_stride,
                                                              float* __restrict__ scale, int64_t rows, int cols,
                                                              float floor_v) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    quant_row_int8(x + row * x_stride, q + row * q_stride, scale + row, cols, floor_v, threadIdx.x & 63);
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
