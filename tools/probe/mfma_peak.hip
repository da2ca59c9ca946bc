// Probe: what the matrix pipe of an MI355X CU sustains for mfma_f32_32x32x16_bf16 / mfma_f32_16x16x32_bf16 under the
// conditions of the grouped-GEMM main loop (512-thread workgroups, 1 per CU, 2 waves per SIMD, 8 independent
// accumulator tiles per wave), and what the other instruction classes of that loop cost next to it.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// VARIANT 0: MFMA only.  1: + 8 cvt per 8 MFMAs (the W8A16 conversion rate: 32 per 32 MFMAs).
//         2: + LDS reads at the loop's rate (6 x b128 per 8 MFMAs), results consumed.  3: 1 + 2.
//         4: 3 + one s_barrier per 32 MFMAs.  6: 3 with the feed interleaved MFMA by MFMA (sched_group_barrier); 5: 6 + barrier.
template <int VARIANT, int WAVES_ACTIVE>
__global__ __launch_bounds__(512, 2) void peak32(int iters, float* out, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = tid >> 6;
    for (int i = tid; i < 64 * 1024 / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + i;
    __syncthreads();
    if (wave >= WAVES_ACTIVE) return;
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    u32x4 araw = {0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    bf16x8 a0 = __builtin_bit_cast(bf16x8, araw), a1 = a0, b0 = a0, b1 = a0;
    unsigned wraw0 = 0x38383838u + lane, wraw1 = 0x40404040u;
    const unsigned char* lp = lds + (lane * 16) + wave * 4096;
    u32x4 n0 = araw, n1 = araw, n2 = araw;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {   // 4 x 8 = 32 MFMAs = one 64-deep stage of the GEMM kernel
            if (VARIANT == 2 || VARIANT >= 3) {   // fragments read one group ahead (software pipelined like the GEMM loop)
                b0 = __builtin_bit_cast(bf16x8, n0 ^ n2);
                b1 = __builtin_bit_cast(bf16x8, n1);
                n0 = *reinterpret_cast<const u32x4*>(lp + ((it + g) & 3) * 1024);
                n1 = *reinterpret_cast<const u32x4*>(lp + 16384 + ((it + g) & 3) * 1024);
                n2 = *reinterpret_cast<const u32x4*>(lp + 32768 + ((it + g) & 3) * 1024);
            }
            if (VARIANT == 1 || VARIANT >= 3) {
                bf16x2 c0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(wraw0, 1.0f, false);
                bf16x2 c1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(wraw0, 1.0f, true);
                bf16x2 c2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(wraw1, 1.0f, false);
                bf16x2 c3 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(wraw1, 1.0f, true);
                a0[0] = c0[0]; a0[1] = c0[1]; a0[2] = c1[0]; a0[3] = c1[1]; a0[4] = c2[0]; a0[5] = c2[1]; a0[6] = c3[0]; a0[7] = c3[1];
                bf16x2 d0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(wraw1, 1.0f, false);
                bf16x2 d1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(wraw1, 1.0f, true);
                bf16x2 d2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(wraw0, 1.0f, false);
                bf16x2 d3 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(wraw0, 1.0f, true);
                a1[0] = d0[0]; a1[1] = d0[1]; a1[2] = d1[0]; a1[3] = d1[1]; a1[4] = d2[0]; a1[5] = d2[1]; a1[6] = d3[0]; a1[7] = d3[1];
                wraw0 += 0x01010101u * (it & 1);
            }
            if (VARIANT < 5) __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[7], 0, 0, 0);
            if (VARIANT < 5) __builtin_amdgcn_sched_barrier(0);
            if (VARIANT >= 5) {
                // issue order inside the group: MFMA, then a slice of the NEXT group's feed (reads / conversions) in its shadow
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                    if (q < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // 2 VALU
                }
            }
        }
        if (VARIANT == 4 || VARIANT == 5) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int VARIANT, int WAVES>
static void run(const char* name, int blocks) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, (size_t)blocks * 512 * 4); hipMalloc(&clk, (size_t)blocks * 16);
    hipMemset(clk, 0, (size_t)blocks * 16);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((peak32<VARIANT, WAVES>), dim3(blocks), dim3(512), 0, 0, iters, out, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL((peak32<VARIANT, WAVES>), dim3(blocks), dim3(512), 0, 0, iters, out, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flop = (double)blocks * WAVES * iters * 32.0 * 2 * 32 * 32 * 16;
    const double cyc_per_mfma_simd = (double)h[0] / (iters * 32.0) / (WAVES / 4.0);   // pipe cycles per MFMA issued on a SIMD
    printf("%-44s %2d waves/CU: %7.1f TFLOP/s  %.3f ms  clock %.3f GHz  %.1f cycles per wave-stage(32 MFMA)  %.1f pipe-cycles/MFMA\n", name, WAVES,
           flop / ms / 1e9, ms, (double)h[0] / h[1] * 0.1, (double)h[0] / iters, cyc_per_mfma_simd);
    hipFree(out); hipFree(clk);
}

int main() {
    int dev_cus = 256;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); dev_cus = prop.multiProcessorCount;
    printf("CUs %d\n", dev_cus);
    run<0, 4>("mfma 32x32x16 only", dev_cus);
    run<0, 8>("mfma 32x32x16 only", dev_cus);
    run<1, 4>("+cvt (8 per 8 MFMA)", dev_cus);
    run<1, 8>("+cvt (8 per 8 MFMA)", dev_cus);
    run<2, 4>("+LDS reads (3 x b128 per 8 MFMA)", dev_cus);
    run<2, 8>("+LDS reads (3 x b128 per 8 MFMA)", dev_cus);
    run<3, 4>("+cvt +LDS", dev_cus);
    run<3, 8>("+cvt +LDS", dev_cus);
    run<4, 8>("+cvt +LDS +barrier per 32 MFMA", dev_cus);
    run<6, 4>("+cvt +LDS, feed interleaved with MFMAs", dev_cus);
    run<6, 8>("+cvt +LDS, feed interleaved with MFMAs", dev_cus);
    run<5, 8>("+cvt +LDS interleaved +barrier", dev_cus);
    return 0;
}
