// Probe: which (operand row, k half) does the scale VGPR of lane l address in v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3)?
// Method: all scales 2^0 except ONE lane's (2^1); the difference to the unit-scale result is the partial product of exactly
// the (row, k-half) that lane's scale applies to.  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int OA, int OB>
__global__ void one_mfma(const unsigned* a, const unsigned* b, const int* sa, const int* sb, float* c) {
    const int l = threadIdx.x;
    i32x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = a[l * 8 + i]; bv[i] = b[l * 8 + i]; }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, OA, sa[l], OB, sb[l]);
    for (int i = 0; i < 16; ++i) c[l * 16 + i] = acc[i];
}
static uint8_t enc_int(int v) {
    if (v == 0) return 0;
    uint8_t s = v < 0 ? 0x80 : 0;
    int a = abs(v), e = 0;
    while ((1 << (e + 1)) <= a) ++e;
    int m = ((a << 3) >> e) & 7;
    return s | (uint8_t)(((e + 7) << 3) | m);
}
static int A[32][64], B[64][32];
static float C0[32][32], C1[32][32];
static void unpack(const float* hc, float C[32][32]) {
    for (int l = 0; l < 64; ++l)
        for (int reg = 0; reg < 16; ++reg) C[(reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5)][l & 31] = hc[l * 16 + reg];
}
int main() {
    srand(11);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) A[i][k] = (rand() % 15) - 7;
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) B[k][j] = (rand() % 13) - 6;
    unsigned ha[64 * 8], hb[64 * 8];
    memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
    for (int l = 0; l < 64; ++l) {
        const int r = l & 31, h = l >> 5;
        for (int j = 0; j < 32; ++j) {
            ha[l * 8 + j / 4] |= (unsigned)enc_int(A[r][32 * h + j]) << (8 * (j & 3));
            hb[l * 8 + j / 4] |= (unsigned)enc_int(B[32 * h + j][r]) << (8 * (j & 3));
        }
    }
    unsigned *da, *db; int *dsa, *dsb; float* dc; float hc[64 * 16]; int hsa[64], hsb[64];
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, sizeof(hc));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    auto run = [&](int opa, int opb) {
        hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
        if (opa == 0 && opb == 0) hipLaunchKernelGGL((one_mfma<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
        else if (opa == 1) hipLaunchKernelGGL((one_mfma<1, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
        else if (opa == 2) hipLaunchKernelGGL((one_mfma<2, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
        else if (opa == 3) hipLaunchKernelGGL((one_mfma<3, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
        else hipLaunchKernelGGL((one_mfma<0, 1>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
        hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
    };
    for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 127;
    run(0, 0);
    unpack(hc, C0);
    for (int which = 0; which < 2; ++which) {
        printf("%s scale: lane -> elements scaled (row/col, k range)\n", which == 0 ? "A" : "B");
        for (int lx = 0; lx < 64; ++lx) {
            for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 127;
            (which == 0 ? hsa : hsb)[lx] = 128;
            run(0, 0);
            unpack(hc, C1);
            // which (row i or col j, k sub-range) explains C1 - C0?  try k ranges of 8 granularity [k0, k1)
            int found = 0;
            for (int idx = 0; idx < 32 && !found; ++idx)
                for (int k0 = 0; k0 < 64 && !found; k0 += 8)
                    for (int k1 = k0 + 8; k1 <= 64 && !found; k1 += 8) {
                        int ok = 1;
                        for (int i = 0; i < 32 && ok; ++i)
                            for (int j = 0; j < 32 && ok; ++j) {
                                double d = 0;
                                if ((which == 0 ? i : j) == idx)
                                    for (int k = k0; k < k1; ++k) d += (double)A[i][k] * B[k][j];
                                if ((double)(C1[i][j] - C0[i][j]) != d) ok = 0;
                            }
                        if (ok) { printf("  lane %2d -> %s %2d, k [%d,%d)\n", lx, which == 0 ? "row" : "col", idx, k0, k1); found = 1; }
                    }
            if (!found) {
                int nz = 0;
                for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) nz += C1[i][j] != C0[i][j];
                printf("  lane %2d -> no single (index, k range) explains it; %d outputs changed\n", lx, nz);
            }
        }
    }
    // opsel: scale byte b of the VGPR selected by opsel = b?
    for (int op = 1; op <= 3; ++op) {
        for (int l = 0; l < 64; ++l) { hsa[l] = 127 | (127 << 8) | (127 << 16) | (127 << 24); hsb[l] = 127; }
        for (int l = 0; l < 64; ++l) hsa[l] = (hsa[l] & ~(0xff << (8 * op))) | (128 << (8 * op));   // byte `op` = x2
        run(op, 0);
        unpack(hc, C1);
        int dbl = 0, same = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { dbl += C1[i][j] == 2 * C0[i][j]; same += C1[i][j] == C0[i][j]; }
        printf("opsel_a = %d with x2 in byte %d of every lane: %d outputs doubled, %d unchanged (of 1024)\n", op, op, dbl, same);
    }
    return 0;
}
