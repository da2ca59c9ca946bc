// Probe: semantics of v_cvt_scalef32_pk_bf16_fp8 on gfx950 (is the fp8->bf16 conversion exact at scale 1.0?
// does the scale use the full f32 value or only its exponent?).  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__global__ void k(const unsigned* in, float scale, unsigned* out) {
    unsigned v = in[threadIdx.x];
    bf16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, scale, false);
    bf16x2 b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, scale, true);
    out[threadIdx.x * 2] = __builtin_bit_cast(unsigned, a);
    out[threadIdx.x * 2 + 1] = __builtin_bit_cast(unsigned, b);
}
static float dec(uint8_t b) {
    int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 15 && m == 7) return NAN;
    v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -v : v;
}
static float bf(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
    unsigned h_in[64], *d_in, *d_out, h_out[128];
    for (int i = 0; i < 64; ++i) h_in[i] = (4 * i) | ((4 * i + 1) << 8) | ((4 * i + 2) << 16) | ((unsigned)(4 * i + 3) << 24);
    hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out));
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    const float scales[] = {1.0f, 2.0f, 1.5f, 0.75f, 0.001f};
    for (float sc : scales) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, sc, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        int bad_exact = 0, bad_exp = 0; int ex; float exp_only = ldexpf(1.f, (frexpf(sc, &ex), ex - 1));
        for (int b = 0; b < 256; ++b) {
            int dw = b / 4, within = b % 4;
            unsigned word = h_out[dw * 2 + within / 2];
            uint16_t hv = within % 2 ? word >> 16 : word & 0xffff;
            float got = bf(hv), want = dec((uint8_t)b);
            if (isnan(want)) { if (!isnan(got)) { bad_exact++; bad_exp++; } continue; }
            // expected with full-scale multiply, rounded to bf16 RNE
            float full = want * sc; uint32_t u; memcpy(&u, &full, 4); u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000u; float fullr; memcpy(&fullr, &u, 4);
            float eo = want * exp_only;
            if (got != fullr) bad_exact++;
            if (got != eo) bad_exp++;
        }
        printf("scale %-6g : mismatches vs full-f32-scale(RNE) = %3d, vs exponent-only-scale = %3d\n", sc, bad_exact, bad_exp);
    }
    return 0;
}
