// Two-term e4m3 split of bf16 activations (see moe_gemm_fp8w_split.hip for why it is exact): shared by the W8A16 split kernel
// and the native MX-fp4 GEMM (gemm_mxfp4.hip).
#pragma once
#include "sglk_common.h"

namespace sglk {

SGLK_DEV int sp_e8m0_for_amax(float amax) {   // as e8m0_for_amax of moe_gemm_a8.hip, floor 5 so that the lo scale (sb - 4) >= 1
    const unsigned u = __float_as_uint(amax);
    int sb = (int)(u >> 23) - 8 + ((u & 0x7fffffu) > 0x600000u ? 1 : 0);
    sb = sb < 5 ? 5 : (sb > 253 ? 253 : sb);
    return sb;
}
SGLK_DEV float sp_pow2(int e) { return __uint_as_float((unsigned)e << 23); }   // 2^(e - 127), 1 <= e <= 254

// (hi, lo) of eight fp32 values with block scale byte sb: two dwords of e4m3 each
SGLK_DEV void split8(const float* v, int sb, unsigned* hi, unsigned* lo) {
    const float inv = sp_pow2(254 - sb), s = sp_pow2(sb), inv_lo = sp_pow2(254 - sb + 4);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int h = 0;
        h = __builtin_amdgcn_cvt_pk_fp8_f32(v[q * 4 + 0] * inv, v[q * 4 + 1] * inv, h, false);
        h = __builtin_amdgcn_cvt_pk_fp8_f32(v[q * 4 + 2] * inv, v[q * 4 + 3] * inv, h, true);
        float r[4];   // the byte selector of the conversion must be a literal
        r[0] = (v[q * 4 + 0] - __builtin_amdgcn_cvt_f32_fp8(h, 0) * s) * inv_lo;
        r[1] = (v[q * 4 + 1] - __builtin_amdgcn_cvt_f32_fp8(h, 1) * s) * inv_lo;
        r[2] = (v[q * 4 + 2] - __builtin_amdgcn_cvt_f32_fp8(h, 2) * s) * inv_lo;
        r[3] = (v[q * 4 + 3] - __builtin_amdgcn_cvt_f32_fp8(h, 3) * s) * inv_lo;
        int l = 0;
        l = __builtin_amdgcn_cvt_pk_fp8_f32(r[0], r[1], l, false);
        l = __builtin_amdgcn_cvt_pk_fp8_f32(r[2], r[3], l, true);
        hi[q] = (unsigned)h;
        lo[q] = (unsigned)l;
    }
}

}  // namespace sglk
