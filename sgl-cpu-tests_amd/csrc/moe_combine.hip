// Top-k reduce of fused_experts: out[m] = sum_j ic2[m*topk + j] over the valid slots, in ascending j, fp32,
// one final bf16 rounding (the routing weight was already applied in the GEMM-2 epilogue).
// Oracle: /root/reference/test_moe_fp8_ext.py:89-91.  Fixed summation order -> run-to-run identical bits.
// HBM-bound: reads topk x K x 2 B and writes K x 2 B per token, 16-byte accesses.
#include "moe_internal.h"

namespace sglk {

__global__ __launch_bounds__(256) void moe_combine_kernel(const uint16_t* __restrict__ ic2,
                                                          const int32_t* __restrict__ topk_ids,
                                                          uint16_t* __restrict__ out, int64_t out_stride, int M, int K,
                                                          int E, int topk) {
    const int chunks_per_row = K >> 3;   // 8 bf16 = 16 B per thread
    const int64_t total = (int64_t)M * chunks_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / chunks_per_row);
        const int c = (int)(i - (int64_t)m * chunks_per_row);
        float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < topk; ++j) {
            const int e = topk_ids[(int64_t)m * topk + j];
            if (e < 0 || e >= E) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(ic2 + ((int64_t)m * topk + j) * K + c * 8);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sum[2 * q] += __uint_as_float(w[q] << 16);
                sum[2 * q + 1] += __uint_as_float(w[q] & 0xffff0000u);
            }
        }
        uint4 o;
        o.x = pack_bf16x2(sum[0], sum[1]);
        o.y = pack_bf16x2(sum[2], sum[3]);
        o.z = pack_bf16x2(sum[4], sum[5]);
        o.w = pack_bf16x2(sum[6], sum[7]);
        *reinterpret_cast<uint4*>(out + (int64_t)m * out_stride + c * 8) = o;
    }
}

// topk known at compile time: all TOPK row loads of a chunk are issued back to back with no branch between them, so
// every thread keeps TOPK 16-byte loads in flight instead of one.  A masked slot (expert id outside [0,E): the -1 of
// expert parallelism, where most slots of a row are masked) re-reads the token's slot-0 address instead of its own
// row -- a cache hit, no new HBM bytes -- and its value is dropped by a select, never added.
template <int TOPK>
__global__ __launch_bounds__(256) void moe_combine_fixed_kernel(const uint16_t* __restrict__ ic2,
                                                                const int32_t* __restrict__ topk_ids,
                                                                uint16_t* __restrict__ out, int64_t out_stride, int M,
                                                                int K, int E) {
    const int chunks_per_row = K >> 3;
    const int64_t total = (int64_t)M * chunks_per_row;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // the expert ids of item i+stride are fetched while item i's rows are in flight, so the id -> row-address
    // dependency is never exposed
    int e_cur[TOPK], e_nxt[TOPK];
    {
        const int m0 = (int)(i / chunks_per_row);
#pragma unroll
        for (int j = 0; j < TOPK; ++j) e_cur[j] = topk_ids[(int64_t)m0 * TOPK + j];
    }
    for (; i < total; i += stride) {
        const int m = (int)(i / chunks_per_row);
        const int c = (int)(i - (int64_t)m * chunks_per_row);
        const uint16_t* base = ic2 + (int64_t)m * TOPK * K + c * 8;
        u32x4 v[TOPK];
        bool valid[TOPK];
#pragma unroll
        for (int j = 0; j < TOPK; ++j) {
            valid[j] = e_cur[j] >= 0 && e_cur[j] < E;
            // row index picked by a select the optimiser cannot turn back into a branch around the load (it does, and
            // then waits for each load before the next branch)
            int jj = valid[j] ? j : 0;
            asm volatile("" : "+v"(jj));
            v[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (int64_t)jj * K));
        }
        {
            const int64_t in = i + stride < total ? i + stride : i;
            const int mn = (int)(in / chunks_per_row);
#pragma unroll
            for (int j = 0; j < TOPK; ++j) e_nxt[j] = topk_ids[(int64_t)mn * TOPK + j];
        }
        float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TOPK; ++j) {
            const bool ok = valid[j];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned w = ok ? v[j][q] : 0u;
                sum[2 * q] += __uint_as_float(w << 16);
                sum[2 * q + 1] += __uint_as_float(w & 0xffff0000u);
            }
        }
        u32x4 o;
        o[0] = pack_bf16x2(sum[0], sum[1]);
        o[1] = pack_bf16x2(sum[2], sum[3]);
        o[2] = pack_bf16x2(sum[4], sum[5]);
        o[3] = pack_bf16x2(sum[6], sum[7]);
        *reinterpret_cast<u32x4*>(out + (int64_t)m * out_stride + c * 8) = o;
#pragma unroll
        for (int j = 0; j < TOPK; ++j) e_cur[j] = e_nxt[j];
    }
}

// any K / alignment: one element per thread (only the odd shapes of the generic path come here)
__global__ __launch_bounds__(256) void moe_combine_scalar_kernel(const uint16_t* __restrict__ ic2,
                                                                 const int32_t* __restrict__ topk_ids,
                                                                 uint16_t* __restrict__ out, int64_t out_stride, int M,
                                                                 int K, int E, int topk) {
    const int64_t total = (int64_t)M * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / K), c = (int)(i - (int64_t)m * K);
        float sum = 0.f;
        for (int j = 0; j < topk; ++j) {
            const int e = topk_ids[(int64_t)m * topk + j];
            if (e < 0 || e >= E) continue;
            sum += bf16_bits_to_f32(ic2[((int64_t)m * topk + j) * K + c]);
        }
        out[(int64_t)m * out_stride + c] = f32_to_bf16_bits(sum);
    }
}

// Expert-parallel combine (sgl_kernel/expert_parallel.py): out[m] = sum over ranks d ascending of rows[table[m][d]]
// (table entry < 0: rank d returned nothing for token m), fp32 sum, one bf16 rounding -- the same fixed order as the
// single-GPU combine, so a run is bit-reproducible.  One 16-byte chunk of one token per thread and iteration.
__global__ __launch_bounds__(256) void ep_reduce_rows_kernel(const uint16_t* __restrict__ rows, int64_t rows_stride,
                                                             const int32_t* __restrict__ table, int G,
                                                             uint16_t* __restrict__ out, int64_t out_stride, int M, int K) {
    const int chunks_per_row = K >> 3;
    const int64_t total = (int64_t)M * chunks_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / chunks_per_row);
        const int c = (int)(i - (int64_t)m * chunks_per_row);
        float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int d = 0; d < G; ++d) {
            const int idx = table[(int64_t)m * G + d];
            if (idx < 0) continue;
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rows + (int64_t)idx * rows_stride + c * 8));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sum[2 * q] += __uint_as_float(v[q] << 16);
                sum[2 * q + 1] += __uint_as_float(v[q] & 0xffff0000u);
            }
        }
        u32x4 o;
        o[0] = pack_bf16x2(sum[0], sum[1]);
        o[1] = pack_bf16x2(sum[2], sum[3]);
        o[2] = pack_bf16x2(sum[4], sum[5]);
        o[3] = pack_bf16x2(sum[6], sum[7]);
        *reinterpret_cast<u32x4*>(out + (int64_t)m * out_stride + c * 8) = o;
    }
}

int launch_moe_combine(const uint16_t* ic2, const int32_t* topk_ids, uint16_t* out, int64_t out_stride, int M,
                       int K, int E, int topk, hipStream_t stream) {
    if (M == 0) return SGLK_OK;
    if (K % 8 != 0 || out_stride % 8 != 0 || ((uintptr_t)out % 16) != 0) {
        int64_t nb = ceil_div((int64_t)M * K, 256);
        if (nb > 256 * 8) nb = 256 * 8;
        hipLaunchKernelGGL(moe_combine_scalar_kernel, dim3((unsigned)nb), dim3(256), 0, stream, ic2, topk_ids, out,
                           out_stride, M, K, E, topk);
        SGLK_CHECK_LAUNCH("moe_combine");
        return SGLK_OK;
    }
    const int64_t total = (int64_t)M * (K >> 3);
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (topk == 8)
        hipLaunchKernelGGL(moe_combine_fixed_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, stream, ic2, topk_ids, out,
                           out_stride, M, K, E);
    else if (topk == 2)
        hipLaunchKernelGGL(moe_combine_fixed_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream, ic2, topk_ids, out,
                           out_stride, M, K, E);
    else
        hipLaunchKernelGGL(moe_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, ic2, topk_ids, out, out_stride,
                           M, K, E, topk);
    SGLK_CHECK_LAUNCH("moe_combine");
    return SGLK_OK;
}

}  // namespace sglk

extern "C" int sglk_ep_reduce_rows(const void* rows, int64_t rows_stride, const int32_t* table, int32_t G, void* out,
                                   int64_t out_stride, int32_t M, int32_t K, void* stream) {
    using namespace sglk;
    SGLK_REQUIRE(M >= 0 && K > 0 && G > 0, SGLK_ERR_INVALID, "ep_reduce_rows: bad sizes M=%d K=%d G=%d", M, K, G);
    if (M == 0) return SGLK_OK;
    SGLK_REQUIRE(rows && table && out, SGLK_ERR_INVALID, "ep_reduce_rows: null pointer");
    SGLK_REQUIRE(K % 8 == 0 && rows_stride % 8 == 0 && out_stride % 8 == 0 && ((uintptr_t)rows % 16) == 0 &&
                     ((uintptr_t)out % 16) == 0, SGLK_ERR_SHAPE, "ep_reduce_rows: rows must be 16-byte aligned, K %% 8 == 0");
    int64_t blocks = ceil_div((int64_t)M * (K >> 3), 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(ep_reduce_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)rows,
                       rows_stride, table, G, (uint16_t*)out, out_stride, M, K);
    SGLK_CHECK_LAUNCH("ep_reduce_rows");
    return SGLK_OK;
}

