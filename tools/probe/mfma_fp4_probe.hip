// Probe: v_mfma_scale_f32_32x32x64_f8f6f4 with an E2M1 (fp4) A operand and an e4m3 B operand -- where do the 32 nibbles of a
// lane sit in k, and which (row, k set) does the A-scale of lane l address?  Build: hipcc --offload-arch=gfx950 -O2.
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
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 4, 0, 0, sa[l], 0, sb[l]);   // A: fp4, B: e4m3
    for (int i = 0; i < 16; ++i) c[l * 16 + i] = acc[i];
}
static uint8_t enc_e4m3(int v) {
    if (v == 0) return 0;
    uint8_t s = v < 0 ? 0x80 : 0;
    int a = abs(v), e = 0;
    while ((1 << (e + 1)) <= a) ++e;
    int m = ((a << 3) >> e) & 7;
    return s | (uint8_t)(((e + 7) << 3) | m);
}
static const int kFp4[8] = {0, 1, 2, 3, 4, 6, 8, 12};      // twice the E2M1 magnitudes 0 .5 1 1.5 2 3 4 6
static int A2[32][64], B[64][32];                          // A2 = 2 * A (integers)
static uint8_t Acode[32][64];
static float C0[32][32], C1[32][32];
static void unpack(const float* hc, float C[32][32]) {
    for (int l = 0; l < 64; ++l)
        for (int reg = 0; reg < 16; ++reg) C[(reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5)][l & 31] = hc[l * 16 + reg];
}
int main() {
    srand(5);
    for (int i = 0; i < 32; ++i)
        for (int k = 0; k < 64; ++k) {
            const int code = rand() % 16;
            Acode[i][k] = (uint8_t)code;
            A2[i][k] = ((code & 8) ? -1 : 1) * kFp4[code & 7];
        }
    for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) B[k][j] = (rand() % 13) - 6;
    unsigned *da, *db; int *dsa, *dsb; float* dc; float hc[64 * 16]; int hsa[64], hsb[64];
    unsigned ha[64 * 8], hb[64 * 8];
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, sizeof(hc));
    memset(hb, 0, sizeof(hb));
    for (int l = 0; l < 64; ++l) {
        const int r = l & 31, h = l >> 5;
        for (int j = 0; j < 32; ++j) hb[l * 8 + j / 4] |= (unsigned)enc_e4m3(B[32 * h + j][r]) << (8 * (j & 3));
    }
    hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    auto run = [&]() {
        hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
        hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dc);
        hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
    };
    for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 127;
    // direct map: ONE nibble (code 2 = 1.0) in one lane; B[k][j] = (k % 16) + 1, then (k / 16) + 1 -> which (row, k) is it?
    {
        static float Ca[32][32], Cb[32][32];
        unsigned hb1[64 * 8], hb2[64 * 8];
        memset(hb1, 0, sizeof(hb1)); memset(hb2, 0, sizeof(hb2));
        for (int l = 0; l < 64; ++l) {
            const int h = l >> 5;
            for (int j = 0; j < 32; ++j) {
                const int k = 32 * h + j;
                hb1[l * 8 + j / 4] |= (unsigned)enc_e4m3((k % 16) + 1) << (8 * (j & 3));
                hb2[l * 8 + j / 4] |= (unsigned)enc_e4m3((k / 16) + 1) << (8 * (j & 3));
            }
        }
        const int lanes[] = {0, 1, 5, 31, 32, 33, 63};
        for (int li = 0; li < 7; ++li) {
            const int l = lanes[li];
            printf("lane %2d (reg.nibble -> row:k):", l);
            for (int nib = 0; nib < 64; ++nib) {          // all 8 registers x 8 nibbles
                memset(ha, 0, sizeof(ha));
                ha[l * 8 + nib / 8] = 2u << (4 * (nib & 7));
                hipMemcpy(db, hb1, sizeof(hb1), hipMemcpyHostToDevice);
                run(); unpack(hc, Ca);
                hipMemcpy(db, hb2, sizeof(hb2), hipMemcpyHostToDevice);
                run(); unpack(hc, Cb);
                int row = -1, cnt = 0;
                for (int i = 0; i < 32; ++i) if (Ca[i][0] != 0.f) { row = i; ++cnt; }
                if (cnt == 0) { if (nib % 8 == 0) printf(" r%d:-", nib / 8); continue; }
                const int k = ((int)Cb[row][0] - 1) * 16 + ((int)Ca[row][0] - 1);
                printf(" %d.%d->%d:%d%s", nib / 8, nib & 7, row, k, cnt > 1 ? "(multi)" : "");
            }
            printf("\n");
        }
        hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    }
    // layout hypotheses: lane (r, h) nibble j <-> k = 32h + j; nibble j in byte j/2, low nibble first (lo = 1) or high first
    int good_lo = -1;
    for (int lo = 1; lo >= 0; --lo) {
        memset(ha, 0, sizeof(ha));
        for (int l = 0; l < 64; ++l) {
            const int r = l & 31, h = l >> 5;
            for (int j = 0; j < 32; ++j) {
                const int sh = 8 * ((j / 2) & 3) + 4 * ((j & 1) ^ (lo ? 0 : 1));
                // nibble j of the lane: chunk j / 16, k (in the labelling of the e4m3 B operand above) = 32 (j / 16) + 16 h + j % 16
                ha[l * 8 + j / 8] |= (unsigned)Acode[r][32 * (j / 16) + 16 * h + (j % 16)] << sh;
            }
        }
        run();
        unpack(hc, C0);
        int bad = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double d = 0;
                for (int k = 0; k < 64; ++k) d += 0.5 * A2[i][k] * B[k][j];
                bad += (double)C0[i][j] != d;
            }
        printf("layout: lane (r,h) nibble j = k 32(j/16)+16h+j%%16, %s nibble first: %d of 1024 outputs differ\n", lo ? "low" : "high", bad);
        if (bad == 0) { good_lo = lo; break; }
    }
    if (good_lo < 0) { printf("no layout hypothesis matched\n"); return 0; }
    // A-scale map: one lane's scale x2; which (row, set of 8-wide k groups) explains the change?
    printf("A scale: lane -> row, k groups of 8 (bit g = k in [8g, 8g+8))\n");
    for (int lx = 0; lx < 64; ++lx) {
        for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 127;
        hsa[lx] = 128;
        run();
        unpack(hc, C1);
        int found = 0;
        for (int idx = 0; idx < 32 && !found; ++idx)
            for (int mask = 1; mask < 256 && !found; ++mask) {
                int ok = 1;
                for (int i = 0; i < 32 && ok; ++i)
                    for (int j = 0; j < 32 && ok; ++j) {
                        double d = 0;
                        if (i == idx)
                            for (int k = 0; k < 64; ++k)
                                if (mask >> (k / 8) & 1) d += 0.5 * A2[i][k] * B[k][j];
                        if ((double)(C1[i][j] - C0[i][j]) != d) ok = 0;
                    }
                if (ok) { printf("  lane %2d -> row %2d, k groups 0x%02x\n", lx, idx, mask); found = 1; }
            }
        if (!found) printf("  lane %2d -> not explained\n", lx);
    }
    // B-scale map with the fp4 A operand
    printf("B scale: lane -> col, k groups\n");
    for (int lx = 0; lx < 64; lx += 9) {
        for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 127;
        hsb[lx] = 128;
        run();
        unpack(hc, C1);
        int found = 0;
        for (int idx = 0; idx < 32 && !found; ++idx)
            for (int mask = 1; mask < 256 && !found; ++mask) {
                int ok = 1;
                for (int i = 0; i < 32 && ok; ++i)
                    for (int j = 0; j < 32 && ok; ++j) {
                        double d = 0;
                        if (j == idx)
                            for (int k = 0; k < 64; ++k)
                                if (mask >> (k / 8) & 1) d += 0.5 * A2[i][k] * B[k][j];
                        if ((double)(C1[i][j] - C0[i][j]) != d) ok = 0;
                    }
                if (ok) { printf("  lane %2d -> col %2d, k groups 0x%02x\n", lx, idx, mask); found = 1; }
            }
        if (!found) printf("  lane %2d -> not explained\n", lx);
    }
    return 0;
}
