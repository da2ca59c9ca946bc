// Probe: can the two-term e4m3 split of fp8_split.h (multiply, v_cvt_pk_fp8_f32, v_cvt_f32_fp8, subtract, multiply, convert) be done
// with the SCALED conversions v_cvt_scalef32_pk_fp8_f32 / v_cvt_scalef32_pk_f32_fp8 (half the VALU instructions)?  Compares the bytes
// of both forms over random bf16 blocks (wide exponent spread, zeros, the block maximum at either end of a binade) and prints what the
// scale operand does (divide on the way down, multiply on the way up; exponent only or full value).
// Build: hipcc --offload-arch=gfx950 -O2 -I sgl-cpu-tests_amd/csrc -I include tools/probe/cvt_split_probe.hip -o tools/probe/_build/cvt_split_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

__device__ int e8m0_for_amax(float amax) {
    const unsigned u = __float_as_uint(amax);
    int sb = (int)(u >> 23) - 8 + ((u & 0x7fffffu) > 0x600000u ? 1 : 0);
    return sb < 5 ? 5 : (sb > 253 ? 253 : sb);
}
__device__ float pow2(int e) { return __uint_as_float((unsigned)e << 23); }

__global__ void k(const unsigned short* x, unsigned* out_ref, unsigned* out_fast, int* sbs) {
    // one thread = 8 values of one 128-block (16 threads per block); amax over the block by shuffles
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float v[8];
    float amax = 0.f;
    for (int i = 0; i < 8; ++i) {
        v[i] = __uint_as_float((unsigned)x[t * 8 + i] << 16);
        amax = fmaxf(amax, fabsf(v[i]));
    }
    for (int m = 1; m < 16; m <<= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
    const int sb = e8m0_for_amax(amax);
    sbs[t] = sb;
    const float inv = pow2(254 - sb), s = pow2(sb), inv_lo = pow2(254 - sb + 4), s_lo = pow2(sb - 4);
    for (int q = 0; q < 2; ++q) {
        int h = 0;
        h = __builtin_amdgcn_cvt_pk_fp8_f32(v[q * 4 + 0] * inv, v[q * 4 + 1] * inv, h, false);
        h = __builtin_amdgcn_cvt_pk_fp8_f32(v[q * 4 + 2] * inv, v[q * 4 + 3] * inv, h, true);
        float r[4];
        r[0] = (v[q * 4 + 0] - __builtin_amdgcn_cvt_f32_fp8(h, 0) * s) * inv_lo;
        r[1] = (v[q * 4 + 1] - __builtin_amdgcn_cvt_f32_fp8(h, 1) * s) * inv_lo;
        r[2] = (v[q * 4 + 2] - __builtin_amdgcn_cvt_f32_fp8(h, 2) * s) * inv_lo;
        r[3] = (v[q * 4 + 3] - __builtin_amdgcn_cvt_f32_fp8(h, 3) * s) * inv_lo;
        int l = 0;
        l = __builtin_amdgcn_cvt_pk_fp8_f32(r[0], r[1], l, false);
        l = __builtin_amdgcn_cvt_pk_fp8_f32(r[2], r[3], l, true);
        out_ref[t * 4 + q * 2] = (unsigned)h;
        out_ref[t * 4 + q * 2 + 1] = (unsigned)l;
        // scaled conversions: down = src / scale, up = src * scale (if the hardware does what the names suggest)
        s16x2 hh = {0, 0};
        hh = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(hh, v[q * 4 + 0], v[q * 4 + 1], s, false);
        hh = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(hh, v[q * 4 + 2], v[q * 4 + 3], s, true);
        const unsigned hw = __builtin_bit_cast(unsigned, hh);
        const f32x2_t b01 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(hw, s, false);
        const f32x2_t b23 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(hw, s, true);
        s16x2 ll = {0, 0};
        ll = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(ll, v[q * 4 + 0] - b01[0], v[q * 4 + 1] - b01[1], s_lo, false);
        ll = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(ll, v[q * 4 + 2] - b23[0], v[q * 4 + 3] - b23[1], s_lo, true);
        out_fast[t * 4 + q * 2] = hw;
        out_fast[t * 4 + q * 2 + 1] = __builtin_bit_cast(unsigned, ll);
    }
}

int main() {
    const int blocks128 = 1 << 15, n = blocks128 * 128, threads = n / 8;
    unsigned short* hx = (unsigned short*)malloc(n * 2);
    srand(7);
    for (int b = 0; b < blocks128; ++b) {
        const int base_e = 90 + rand() % 70;          // block level: 2^-37 .. 2^32
        const int spread = 1 + rand() % 24;           // exponent spread inside the block
        for (int i = 0; i < 128; ++i) {
            const int kind = rand() % 50;
            unsigned short v;
            if (kind == 0) v = 0;                                            // zero
            else if (kind == 1) v = (unsigned short)(rand() & 0x807f);       // bf16 denormal
            else {
                const int e = base_e - rand() % spread;
                v = (unsigned short)(((rand() & 1) << 15) | ((e & 0xff) << 7) | (rand() & 0x7f));
            }
            hx[b * 128 + i] = v;
        }
        if (b % 3 == 0) hx[b * 128 + rand() % 128] = (unsigned short)((base_e << 7) | 0x7f);   // maximum at the top of a binade
        if (b % 3 == 1) hx[b * 128 + rand() % 128] = (unsigned short)(((base_e + 1) << 7) | 0x00);   // ... at the bottom of the next
    }
    unsigned short* dx; unsigned *dr, *df; int* ds;
    hipMalloc(&dx, n * 2); hipMalloc(&dr, threads * 16); hipMalloc(&df, threads * 16); hipMalloc(&ds, threads * 4);
    hipMemcpy(dx, hx, n * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(threads / 256), dim3(256), 0, 0, dx, dr, df, ds);
    unsigned* hr = (unsigned*)malloc(threads * 16); unsigned* hf = (unsigned*)malloc(threads * 16);
    hipMemcpy(hr, dr, threads * 16, hipMemcpyDeviceToHost);
    hipMemcpy(hf, df, threads * 16, hipMemcpyDeviceToHost);
    long bad_hi = 0, bad_lo = 0;
    for (int t = 0; t < threads; ++t)
        for (int q = 0; q < 2; ++q) {
            const unsigned a = hr[t * 4 + q * 2] ^ hf[t * 4 + q * 2], b = hr[t * 4 + q * 2 + 1] ^ hf[t * 4 + q * 2 + 1];
            for (int j = 0; j < 4; ++j) { bad_hi += ((a >> (8 * j)) & 0xff) != 0; bad_lo += ((b >> (8 * j)) & 0xff) != 0; }
        }
    printf("values %d: hi bytes that differ %ld, lo bytes that differ %ld (0 / 0 = the scaled conversions reproduce fp8_split.h)\n", n, bad_hi, bad_lo);
    return bad_hi || bad_lo;
}
