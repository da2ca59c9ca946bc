// Probe: operand layout and scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands on gfx950, and its
// issue rate.  Build: hipcc --offload-arch=gfx950 -O2 -o _build/mfma_scale_probe mfma_scale_probe.hip
//
// Hypothesis H (checked with exact small-integer data, asymmetric A and B):
//   A operand, lane l (r = l & 31, h = l >> 5): byte j (0..31, little-endian over the 8 VGPRs) = A[row r][k = 32 h + j]
//   B operand, lane l:                          byte j                                          = B[k = 32 h + j][col r]
//   C/D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)         (the 32x32 map of every dtype)
//   scale_a / scale_b: byte 0 of the lane's scale VGPR (opsel 0) = E8M0 exponent applied to that lane's 32 values
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void one_mfma(const unsigned* a, const unsigned* b, const int* sa, const int* sb, float* c) {
    const int l = threadIdx.x;
    i32x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = a[l * 8 + i]; bv[i] = b[l * 8 + i]; }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, sa[l], 0, sb[l]);
    for (int i = 0; i < 16; ++i) c[l * 16 + i] = acc[i];
}

// issue rate: N back-to-back scaled MFMAs on 4 independent accumulators, one wave per SIMD (256 threads)
__global__ void rate(float* out, int iters, unsigned long long* cyc) {
    i32x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = 0x38383838 + threadIdx.x * 0x01010101 * (i & 1); bv[i] = 0x30303030 + i; }
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[t], 0, 0, 0, 127, 0, 127);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static uint8_t enc_int(int v) {   // exact e4m3 encoding of an integer in [-16, 16]
    if (v == 0) return 0;
    uint8_t s = v < 0 ? 0x80 : 0;
    int a = abs(v), e = 0;
    while ((1 << (e + 1)) <= a) ++e;          // a in [2^e, 2^(e+1))
    int m = ((a << 3) >> e) & 7;              // 3 mantissa bits (exact for a <= 16)
    return s | (uint8_t)(((e + 7) << 3) | m);
}

int main() {
    static int A[32][64], B[64][32];
    srand(7);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) A[i][k] = (rand() % 17) - 8;
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) B[k][j] = (rand() % 13) - 6 + (j == 5 ? 1 : 0);
    unsigned ha[64 * 8], hb[64 * 8];
    int hsa[64], hsb[64];
    memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
    for (int l = 0; l < 64; ++l) {
        const int r = l & 31, h = l >> 5;
        for (int j = 0; j < 32; ++j) {
            ha[l * 8 + j / 4] |= (unsigned)enc_int(A[r][32 * h + j]) << (8 * (j & 3));
            hb[l * 8 + j / 4] |= (unsigned)enc_int(B[32 * h + j][r]) << (8 * (j & 3));
        }
    }
    unsigned *da, *db; int *dsa, *dsb; float* dc; float hc[64 * 16];
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dsa, sizeof(hsa)); hipMalloc(&dsb, sizeof(hsb)); hipMalloc(&dc, sizeof(hc));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    for (int test = 0; test < 4; ++test) {
        // test 0: unit scales.  1: A rows scaled per (row, k-half) lane.  2: B cols scaled per lane.
        // 3: scale value in byte 1 of the VGPR, byte 0 = 127 (must NOT apply with opsel 0); upper bits garbage
        for (int l = 0; l < 64; ++l) { hsa[l] = 127; hsb[l] = 127; }
        if (test == 1) for (int l = 0; l < 64; ++l) hsa[l] = 127 + ((l * 5) % 7) - 3;
        if (test == 2) for (int l = 0; l < 64; ++l) hsb[l] = 127 + ((l * 3) % 5) - 2;
        if (test == 3) for (int l = 0; l < 64; ++l) { hsa[l] = 127 | (130 << 8) | (0x5a << 16); hsb[l] = 127 | (125 << 8); }
        hipMemcpy(dsa, hsa, sizeof(hsa), hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, sizeof(hsb), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
        hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int reg = 0; reg < 16; ++reg) {
                const int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
                double want = 0;
                for (int k = 0; k < 64; ++k) {
                    const int h = k >> 5;
                    const double sa = ldexp(1.0, (hsa[row + 32 * h] & 0xff) - 127), sb = ldexp(1.0, (hsb[col + 32 * h] & 0xff) - 127);
                    want += (double)A[row][k] * sa * (double)B[k][col] * sb;
                }
                if ((double)hc[l * 16 + reg] != want) {
                    if (bad < 4) printf("  test %d mismatch C[%d][%d]: got %g want %g\n", test, row, col, hc[l * 16 + reg], want);
                    ++bad;
                }
            }
        printf("mfma_scale_f32_32x32x64 e4m3, test %d: %d / 1024 mismatches vs hypothesis H\n", test, bad);
    }
    // issue rate
    float* dout; unsigned long long* dcyc; unsigned long long hcyc = 0;
    hipMalloc(&dout, 256 * 256 * 4); hipMalloc(&dcyc, 8);
    const int iters = 4096;
    hipLaunchKernelGGL(rate, dim3(256), dim3(256), 0, 0, dout, iters, dcyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate, dim3(256), dim3(256), 0, 0, dout, iters, dcyc);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&hcyc, dcyc, 8, hipMemcpyDeviceToHost);
    const double flop = 256.0 * 4 * iters * 4 * (2.0 * 32 * 32 * 64);
    printf("scaled e4m3 MFMA 32x32x64: %.1f shader cycles per MFMA per SIMD (one wave), %.0f TFLOP/s chip-wide bare loop (%.3f ms)\n",
           (double)hcyc / (iters * 4.0), flop / (ms * 1e-3) / 1e12, ms);
    return 0;
}
