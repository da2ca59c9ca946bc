// Shared host/device helpers for the sglk HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sglk.h"

// Developer-only scaffolding -- timing ablations (WRONG results by design) and in-kernel time stamps -- exists only in
// SGLK_DEV_ABLATE builds (`SGLK_DEV_ABLATE=1 python sgl-cpu-tests_amd/build.py` -> libsglk_dev.so, loaded through SGLK_LIB_PATH by
// tools/tile_timeline.py).  In the product build every SGLK_ABL(bits, mask) is the literal `false` and SGLK_PP_ABL is 0: the
// guarded statements compile to nothing and no ablation instantiation is reachable.
#ifdef SGLK_DEV_ABLATE
#define SGLK_ABL(bits, mask) ((((int)(bits)) & (mask)) != 0)
#else
#define SGLK_ABL(bits, mask) false
#endif
#if defined(SGLK_DEV_ABLATE) && defined(SGLK_PP_ABLATE)   // attention.hip: -DSGLK_PP_ABLATE=bits in a developer build
#define SGLK_PP_ABL SGLK_PP_ABLATE
#else
#define SGLK_PP_ABL 0
#endif

namespace sglk {

// ---- error plumbing ---------------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define SGLK_FAIL(code, ...)          \
    do {                              \
        ::sglk::set_error(__VA_ARGS__); \
        return (code);                \
    } while (0)

#define SGLK_REQUIRE(cond, code, ...) \
    do {                              \
        if (!(cond)) SGLK_FAIL(code, __VA_ARGS__); \
    } while (0)

#define SGLK_CHECK_LAUNCH(what)                                                          \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) SGLK_FAIL(SGLK_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device-side vector types -------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define SGLK_DEV __device__ __forceinline__

SGLK_DEV float bf16_bits_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// round-to-nearest-even f32 -> bf16 bits (plain cast keeps NaN a NaN; hipcc emits v_cvt_pk_bf16_f32)
SGLK_DEV unsigned short f32_to_bf16_bits(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
SGLK_DEV unsigned pack_bf16x2(float lo, float hi) {
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

// 16 bytes of a weight stream.  NT: the non-temporal policy, for weights that one workgroup reads once per call (MoE experts at decode
// sizes): same-box A/B of fused_experts fp8, replayed and with rotating weight copies alike (profiles/r03_ab_nt_weights.txt): 7-10 % less
// time from 16 tokens on, nothing at 4, 13 % MORE at one token -- the launcher decides (MoeGemmParams::w_nt).  The bf16 mid kernel's
// dword loads of VNNI-packed weights got 7-12 % SLOWER with it (same file): that kernel keeps the default policy.
template <bool NT>
SGLK_DEV u32x4 ld_stream16(const void* p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    else return *reinterpret_cast<const u32x4*>(p);
}

// g * sigmoid(g) with the hardware exp2 / rcp (1 ulp each): far inside the bf16 rounding of the result
SGLK_DEV float silu_f32(float g) {
    return g * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(g * -1.44269504088896340736f));
}

// 8 XCDs, blocks are dealt round-robin: give every XCD a contiguous range of logical ids (bijective for any n)
SGLK_DEV int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

}  // namespace sglk
