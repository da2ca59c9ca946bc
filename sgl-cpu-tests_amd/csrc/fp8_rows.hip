// Row passes in front of the fp8 matrix-core kernels (moe_gemm_fp8w_s128.hip): one wave per row of bf16 activations.
//   split_fp8_block128 : the two-term e4m3 split (x = hi + lo exactly, fp8_split.h) -> [hi 64 | lo 64] bytes per 64-wide k group
//                        in the packed weight tile's k order + one E8M0 byte per 128-wide block          (W8A16, exact products)
//   quant_fp8_block128 : the opt-in a8 mode's quantiser -> 64 e4m3 bytes per k group, same order, same scale format
// Inside fused_experts both ride in moe_align's second launch (SplitJob); these launches serve the dense GEMMs, the small inputs
// whose align is a single launch, and the C-ABI entry points sglk_split_fp8_block128 / sglk_quant_fp8_block128.
#include "fp8_split.h"
#include "moe_internal.h"

namespace sglk {

__global__ __launch_bounds__(256) void split_fp8_block128_kernel(const uint16_t* __restrict__ x, int64_t x_stride,
                                                                 uint8_t* __restrict__ q, int64_t q_stride,
                                                                 uint8_t* __restrict__ s, int64_t s_stride, int64_t rows,
                                                                 int cols) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    split_row_block128(x + row * x_stride, q + row * q_stride, s + row * s_stride, cols, threadIdx.x & 63);
}

int launch_split_fp8_block128(const uint16_t* x, int64_t x_stride, uint8_t* q, int64_t q_stride, uint8_t* s, int64_t s_stride,
                              int64_t rows, int cols, hipStream_t stream) {
    if (rows == 0) return SGLK_OK;
    if (cols % 128 != 0 || x_stride % 8 != 0 || ((uintptr_t)x % 16) != 0 || q_stride % 16 != 0 || ((uintptr_t)q % 16) != 0)
        SGLK_FAIL(SGLK_ERR_SHAPE, "split_fp8_block128: %d columns / strides / alignment not supported", cols);
    hipLaunchKernelGGL(split_fp8_block128_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, stream, x, x_stride, q, q_stride,
                       s, s_stride, rows, cols);
    SGLK_CHECK_LAUNCH("split_fp8_block128");
    return SGLK_OK;
}

__global__ __launch_bounds__(256) void quant_fp8_block128_kernel(const uint16_t* __restrict__ x, int64_t x_stride,
                                                                 uint8_t* __restrict__ q, int64_t q_stride,
                                                                 uint8_t* __restrict__ s, int64_t s_stride, int64_t rows,
                                                                 int cols) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    quant_row_block128(x + row * x_stride, q + row * q_stride, s + row * s_stride, cols, threadIdx.x & 63);
}

int launch_quant_fp8_block128(const uint16_t* x, int64_t x_stride, uint8_t* q, int64_t q_stride, uint8_t* s, int64_t s_stride,
                              int64_t rows, int cols, hipStream_t stream) {
    if (rows == 0) return SGLK_OK;
    if (cols % 128 != 0 || x_stride % 8 != 0 || ((uintptr_t)x % 16) != 0 || q_stride % 16 != 0 || ((uintptr_t)q % 16) != 0)
        SGLK_FAIL(SGLK_ERR_SHAPE, "quant_fp8_block128: %d columns / strides / alignment not supported", cols);
    hipLaunchKernelGGL(quant_fp8_block128_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, stream, x, x_stride, q, q_stride,
                       s, s_stride, rows, cols);
    SGLK_CHECK_LAUNCH("quant_fp8_block128");
    return SGLK_OK;
}

}  // namespace sglk
