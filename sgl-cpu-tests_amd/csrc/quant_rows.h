// Per-row symmetric int8 quantisation of one bf16 row by ONE wave (shared by quant.hip's kernel and the extra workgroups of
// moe_align's placing launch, fp8_split.h: SplitJob).
// Oracle: /root/reference/test_gemm_int8.py:14-22 (floor 1e-10), /root/reference/test_moe_int8.py:23-31 (floor 1e-7):
//   amax = max(|row|, floor); scale = amax / 127; q = round_half_even(x * (127 / amax))
#pragma once
#include "sglk_common.h"

namespace sglk {

SGLK_DEV void quant_row_int8(const uint16_t* __restrict__ xr, int8_t* __restrict__ qr, float* __restrict__ scale_out, int cols,
                             float floor_v, int lane) {
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
    if (lane == 0) *scale_out = amax / 127.0f;
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

}  // namespace sglk
