// Two-term e4m3 split of bf16 activations: shared by the W8A16 kernels on the block-scaled fp8 matrix cores
// (moe_gemm_fp8w_s128.hip) and the native MX-fp4 GEMM (gemm_mxfp4.hip).
//
// Why: v_mfma_scale_f32_32x32x64_f8f6f4 multiplies fp8 x fp8 at twice the bf16 rate and takes fp8 weights AS THEY ARE (no
// fp8 -> bf16 conversion in the loop).  A bf16 activation has 8 significant bits, an e4m3 value 4: x = hi + lo with
//     hi = e4m3(x / s),   lo = e4m3((x - hi * s) / (s / 16)),   s = 2^sb the block's power-of-two scale (amax / s <= 448)
// is EXACT for every element within 2^13 of its 128-block's largest magnitude (hi is then a normal e4m3 number; the residual of
// the first rounding has at most four significant bits and sits at most 2^-4 below the element); smaller elements lose bits
// below 2^-21 * amax(block) -- four orders of magnitude under the bf16 rounding of the result.  Two scaled MFMAs (hi, lo) per
// 64-wide k group cost the matrix pipe exactly what the four bf16 MFMAs of the same k range cost; products stay exact,
// accumulation fp32, the weights' block scales exact (power of two in the instruction, mantissa by the accumulator-unit trick).
// A split row stores, for every 64-wide k group, [hi 64 B | lo 64 B] in the k order of the packed weight tile = 128 contiguous
// bytes per token and stage (one cache line); one E8M0 byte per token and 128-wide block.
#pragma once
#include "sglk_common.h"

namespace sglk {

// E8M0 byte of the power-of-two scale of a 128-wide block with largest magnitude `amax` (>= 0): the smallest 2^e with
// amax / 2^e <= 448 = 1.75 * 2^8 (e4m3's largest finite value), clamped to [1, 253].  Integer arithmetic on the float's
// bits, so the oracle (oracle/moe_a8.py: e8m0_for_amax) reproduces it exactly.
SGLK_DEV int e8m0_for_amax(float amax) {
    const unsigned u = __float_as_uint(amax);
    int sb = (int)(u >> 23) - 8 + ((u & 0x7fffffu) > 0x600000u ? 1 : 0);
    sb = sb < 1 ? 1 : (sb > 253 ? 253 : sb);
    return sb;
}
SGLK_DEV float inv_scale_of(int sb) { return __uint_as_float((unsigned)(254 - sb) << 23); }   // 2^(127 - sb), exact


SGLK_DEV int sp_e8m0_for_amax(float amax) {   // as e8m0_for_amax above, floor 5 so that the lo scale (sb - 4) >= 1
    const unsigned u = __float_as_uint(amax);
    int sb = (int)(u >> 23) - 8 + ((u & 0x7fffffu) > 0x600000u ? 1 : 0);
    sb = sb < 5 ? 5 : (sb > 253 ? 253 : sb);
    return sb;
}
SGLK_DEV float sp_pow2(int e) { return __uint_as_float((unsigned)e << 23); }   // 2^(e - 127), 1 <= e <= 254

// (hi, lo) of eight fp32 values with block scale byte sb: two dwords of e4m3 each.  hi = e4m3(v / s), lo = e4m3((v - hi * s) /
// (s / 16)) with the SCALED conversions (v_cvt_scalef32_pk_fp8_f32 divides by the scale on the way down,
// v_cvt_scalef32_pk_f32_fp8 multiplies on the way up; s a power of two): 2.5 VALU instructions per value instead of 5, and
// byte for byte what the multiply / convert / subtract / multiply / convert form gave (tools/probe/cvt_split_probe.hip: 0
// differences over 4 M values incl. zeros, denormals, 24-binade blocks; profiles/r03_cvt_split_probe.txt)
SGLK_DEV void split8(const float* v, int sb, unsigned* hi, unsigned* lo) {
    typedef __attribute__((ext_vector_type(2))) short s16x2;
    const float s = sp_pow2(sb), s_lo = sp_pow2(sb - 4);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        s16x2 h = {0, 0};
        h = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(h, v[q * 4 + 0], v[q * 4 + 1], s, false);
        h = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(h, v[q * 4 + 2], v[q * 4 + 3], s, true);
        const unsigned hw = __builtin_bit_cast(unsigned, h);
        const f32x2 b01 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(hw, s, false);
        const f32x2 b23 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(hw, s, true);
        s16x2 l = {0, 0};
        l = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(l, v[q * 4 + 0] - b01[0], v[q * 4 + 1] - b01[1], s_lo, false);
        l = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(l, v[q * 4 + 2] - b23[0], v[q * 4 + 3] - b23[1], s_lo, true);
        hi[q] = hw;
        lo[q] = __builtin_bit_cast(unsigned, l);
    }
}

// One row of bf16 activations -> its split row, by ONE wave: q = [hi 64 | lo 64] bytes per 64-wide k group in the packed weight
// tile's k order (octet o = k / 8 of the group sits at byte 16 * (o & 3) + 8 * (o >> 2) of the hi / lo half), s = one E8M0 byte
// per 128-wide block.  A lane takes the two octets o' and o' + 4 of a group (k = 8 o' .. and 32 + 8 o' ..): their hi bytes are
// 16 contiguous output bytes, so a pass of 1024 columns is two 16-byte loads and two 16-byte stores per lane, four lanes
// per group (64 contiguous bytes in and out); the block's amax is a three-step exchange inside 8 lanes.
SGLK_DEV void split_row_block128(const uint16_t* __restrict__ xr, uint8_t* __restrict__ qr, uint8_t* __restrict__ sr, int cols,
                                 int lane) {
    for (int c0 = 0; c0 < cols; c0 += 1024) {
        const int g = lane >> 2, o = lane & 3;
        const int c = c0 + g * 64;
        const bool live = c < cols;
        float v[16];
        float amax = 0.f;
        if (live) {
            const uint4 a4 = *reinterpret_cast<const uint4*>(xr + c + 8 * o);
            const uint4 b4 = *reinterpret_cast<const uint4*>(xr + c + 32 + 8 * o);
            const unsigned w[8] = {a4.x, a4.y, a4.z, a4.w, b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[2 * i] = __uint_as_float(w[i] << 16);
                v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) amax = fmaxf(amax, fabsf(v[j]));
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = 0.f;
        }
        amax = fmaxf(amax, __shfl_xor(amax, 1));
        amax = fmaxf(amax, __shfl_xor(amax, 2));
        amax = fmaxf(amax, __shfl_xor(amax, 4));
        const int sb = sp_e8m0_for_amax(amax);
        if (live) {
            if ((lane & 7) == 0) sr[c >> 7] = (uint8_t)sb;
            unsigned h0[2], l0[2], h1[2], l1[2];
            split8(v, sb, h0, l0);
            split8(v + 8, sb, h1, l1);
            uint8_t* g64 = qr + 2 * c + 16 * o;
            *reinterpret_cast<uint4*>(g64) = make_uint4(h0[0], h0[1], h1[0], h1[1]);
            *reinterpret_cast<uint4*>(g64 + 64) = make_uint4(l0[0], l0[1], l1[0], l1[1]);
        }
    }
}

// The a8 mode's quantiser for one row, same lane mapping: q = 64 e4m3 bytes per 64-wide k group in the packed weight tile's k
// order, the smallest power-of-two scale with amax / 2^e <= 448 per 128-wide block (e8m0_for_amax: floor 1, not 5)
SGLK_DEV void quant_row_block128(const uint16_t* __restrict__ xr, uint8_t* __restrict__ qr, uint8_t* __restrict__ sr, int cols,
                                 int lane) {
    for (int c0 = 0; c0 < cols; c0 += 1024) {
        const int g = lane >> 2, o = lane & 3;
        const int c = c0 + g * 64;
        const bool live = c < cols;
        float v[16];
        float amax = 0.f;
        if (live) {
            const uint4 a4 = *reinterpret_cast<const uint4*>(xr + c + 8 * o);
            const uint4 b4 = *reinterpret_cast<const uint4*>(xr + c + 32 + 8 * o);
            const unsigned w[8] = {a4.x, a4.y, a4.z, a4.w, b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[2 * i] = __uint_as_float(w[i] << 16);
                v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) amax = fmaxf(amax, fabsf(v[j]));
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = 0.f;
        }
        amax = fmaxf(amax, __shfl_xor(amax, 1));
        amax = fmaxf(amax, __shfl_xor(amax, 2));
        amax = fmaxf(amax, __shfl_xor(amax, 4));
        const int sb = e8m0_for_amax(amax);
        const float inv = inv_scale_of(sb);
        if (live) {
            if ((lane & 7) == 0) sr[c >> 7] = (uint8_t)sb;
            int d[4] = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d[i] = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * i + 0] * inv, v[4 * i + 1] * inv, d[i], false);
                d[i] = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * i + 2] * inv, v[4 * i + 3] * inv, d[i], true);
            }
            *reinterpret_cast<uint4*>(qr + c + 16 * o) = make_uint4((unsigned)d[0], (unsigned)d[1], (unsigned)d[2], (unsigned)d[3]);
        }
    }
}

// the split of `hidden` as extra workgroups of another launch (moe_align's second kernel: the two are independent and
// together shorter than back to back)
struct SplitJob {
    const uint16_t* x;
    int64_t x_stride;       // elements
    uint8_t* q;
    int64_t q_stride;       // bytes (>= 2 * cols)
    uint8_t* s;
    int64_t s_stride;       // bytes
    int64_t rows;
    int cols;
    int terms;              // 2: the two-term split (q rows of 2 * cols bytes); 1: the a8 mode's quantised rows (cols bytes);
                            // 0: the W8A8 operator's per-token int8 rows (quant_rows.h), one f32 factor per row in sf
    float* sf;
    float floor_v;
};

}  // namespace sglk
