// mxfp4_scaled_mm_cpu on the block-scaled matrix cores with the weights AS STORED (/root/reference/test_mxfp4.py:148-204;
// oracle /root/reference/test_mxfp4.py:14-127 restated in oracle/gemm.py).
//
// v_mfma_scale_f32_32x32x64_f8f6f4 takes an E2M1 (fp4) A operand with one E8M0 scale per 32 consecutive k -- exactly the MX
// block structure of the weight -- and an e4m3 B operand.  The reference contract is W4A16 (activations are never quantised),
// so the bf16 activations go in as TWO e4m3 terms (hi + lo under a power-of-two block scale, fp8_split.h: exact for every
// element within 2^13 of its 128-block's maximum; smaller ones lose bits 2^-21 below that maximum), two MFMAs per 64-deep
// k-step: every product is exact, accumulation is fp32, one bf16 rounding of the result.  No dequantised copy of the weight
// ever exists; per 64 k a wave reads 16 bytes of weight per row.
//
// Operand order, measured with tools/probe/mfma_fp4_probe.hip / mfma_scale_map.hip (profiles/r02_mfma_fp4_probe.txt), lane =
// (r = lane & 31, h = lane >> 5):
//   fp4 A operand : the lane's 32 nibbles (four registers, low nibble first) are k = 32h .. 32h+31 of the 64-deep step -- ONE MX
//                   block, 16 contiguous bytes of the reference's row-major nibble matrix -- and the lane's own scale register
//                   applies to them: the weight and its E8M0 scales go in exactly as stored;
//   e4m3 B operand: bytes 0-15 are k = 16h .. 16h+15, bytes 16-31 are k = 32+16h .. 32+16h+15; the scale register of lanes
//                   0-31 applies to k 0..31 and that of lanes 32-63 to k 32..63 (both inside one 128-wide activation block).
//
// Shape: 128 weight rows x 128 tokens per workgroup; four waves, each 128 rows x 32 tokens (four 32x32 accumulators), all
// operands straight from global memory / L2 into registers (an e4m3 row's 64-byte line per k-step is consumed whole by the two
// lanes of a token; the fp4 rows are tiny), one k-step of loads in flight ahead of the MFMAs.  Used for M >= 64, N % 128 == 0,
// K % 256 == 0 when the 128 x 128 tiles fill at least half the chip; other shapes stay on the generic engine's exact bf16 expansion.
#include "fp8_split.h"
#include "knobs.h"
#include "moe_internal.h"

namespace sglk {
namespace gfp4 {

typedef __attribute__((ext_vector_type(8))) int i32x8;

// x [rows][cols] bf16 -> q [rows][hi cols | lo cols] e4m3 in natural k order + one E8M0 byte per 128-wide block.
// One wave per row and 2048 columns per pass (lane = 32 consecutive columns, 4 lanes = one block).
__global__ __launch_bounds__(256) void split_natural_kernel(const uint16_t* __restrict__ x, int64_t x_stride, uint8_t* __restrict__ q,
                                                            int64_t q_stride, uint8_t* __restrict__ s, int64_t s_stride, int64_t rows,
                                                            int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const uint16_t* xr = x + row * x_stride;
    uint8_t* qr = q + row * q_stride;
    for (int c0 = 0; c0 < cols; c0 += 2048) {
        const int c = c0 + lane * 32;
        const bool live = c < cols;
        float v[32];
        float amax = 0.f;
        if (live) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4 w4 = *reinterpret_cast<const uint4*>(xr + c + j * 8);
                const unsigned w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[j * 8 + 2 * i] = __uint_as_float(w[i] << 16);
                    v[j * 8 + 2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
                }
            }
#pragma unroll
            for (int j = 0; j < 32; ++j) amax = fmaxf(amax, fabsf(v[j]));
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) v[j] = 0.f;
        }
        amax = fmaxf(amax, __shfl_xor(amax, 1));
        amax = fmaxf(amax, __shfl_xor(amax, 2));
        const int sb = sp_e8m0_for_amax(amax);
        if (live) {
            if ((lane & 3) == 0) s[row * s_stride + (c >> 7)] = (uint8_t)sb;
            unsigned hi[8], lo[8];
#pragma unroll
            for (int run = 0; run < 4; ++run) split8(v + run * 8, sb, hi + 2 * run, lo + 2 * run);
            *reinterpret_cast<uint4*>(qr + c) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            *reinterpret_cast<uint4*>(qr + c + 16) = make_uint4(hi[4], hi[5], hi[6], hi[7]);
            *reinterpret_cast<uint4*>(qr + cols + c) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            *reinterpret_cast<uint4*>(qr + cols + c + 16) = make_uint4(lo[4], lo[5], lo[6], lo[7]);
        }
    }
}

struct Fp4Params {
    const uint8_t* xq;     // [M][2K]: hi | lo
    int64_t xq_stride;
    const uint8_t* xs;     // [M][K/128] E8M0
    int64_t xs_stride;
    const uint8_t* wq;     // [N][K/2]
    const uint8_t* ws;     // E8M0 per (row, 32 k): [N][K/32] or the packed order [N/32][K/32][32]
    int scale_packed;
    const float* bias;
    uint16_t* out;
    int64_t out_stride;
    int M, N, K;
};

template <int RT>
struct StepRegs {          // the operands of one 64-deep k-step of a wave
    uint4 a[RT];           // the lane's MX block of each of the wave's row tiles
    uint4 bh[2], bl[2];    // activation chunks, hi and lo term
};

// RT = 32-row tiles per wave (the workgroup's weight rows): 4 -> 192 registers, two waves per SIMD; 2 -> four waves per SIMD, more
// loads in flight per SIMD for half the reuse of an activation fragment
template <int RT>
__global__ __launch_bounds__(256) void gemm_mxfp4_kernel(const Fp4Params p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * (RT * 32);
    const int tok = blockIdx.y * 128 + wave * 32 + r;
    const int tokc = tok < p.M ? tok : p.M - 1;             // rows past M re-read the last one, never stored
    const int K = p.K, KB = K >> 5;
    const uint8_t* xhi = p.xq + (int64_t)tokc * p.xq_stride + 16 * h;
    const uint8_t* xlo = xhi + K;
    const uint8_t* xsr = p.xs + (int64_t)tokc * p.xs_stride;
    const uint8_t* wrow[RT];
    const uint8_t* srow[RT];                                 // this lane's scale of 32-block kb sits at srow[rt][kb * sstep]
    const int sstep = p.scale_packed ? 32 : 1;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int n = n0 + rt * 32 + r;
        wrow[rt] = p.wq + (int64_t)n * (K >> 1) + 16 * h;
        srow[rt] = p.scale_packed ? p.ws + ((int64_t)(n >> 5) * KB) * 32 + (n & 31) : p.ws + (int64_t)n * KB;
    }
    f32x16 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[rt][i] = 0.f;

    auto load_step = [&](int t, StepRegs<RT>& g) __attribute__((always_inline)) {
        const int kb = t * 32;                               // byte offset of the step inside a weight row
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) g.a[rt] = *reinterpret_cast<const uint4*>(wrow[rt] + kb);
        const int k0 = t * 64;
        g.bh[0] = *reinterpret_cast<const uint4*>(xhi + k0);
        g.bh[1] = *reinterpret_cast<const uint4*>(xhi + k0 + 32);
        g.bl[0] = *reinterpret_cast<const uint4*>(xlo + k0);
        g.bl[1] = *reinterpret_cast<const uint4*>(xlo + k0 + 32);
    };
    // scales of the four k-steps of a 256-wide group: byte s of sa[rt] = MX block 8 q + 2 s + h of the lane's row;
    // byte s of sxh = the activation block scale of k-step s (one per 128), sxl = the lo term's (4 binades below)
    auto load_scales = [&](int q, unsigned (&sa)[RT], unsigned& sxh, unsigned& sxl) __attribute__((always_inline)) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const uint8_t* sp = srow[rt] + (int64_t)(8 * q + h) * sstep;
            sa[rt] = (unsigned)sp[0] | ((unsigned)sp[2 * sstep] << 8) | ((unsigned)sp[4 * sstep] << 16) | ((unsigned)sp[6 * sstep] << 24);
        }
        const unsigned b0 = xsr[2 * q], b1 = xsr[2 * q + 1];
        sxh = b0 | (b0 << 8) | (b1 << 16) | (b1 << 24);
        sxl = sxh - 0x04040404u;                             // every byte >= 5 (sp_e8m0_for_amax)
    };
#define SGLK_FP4_STEP(S, G)                                                                                                  \
    {                                                                                                                        \
        const i32x8 bh = {(int)G.bh[0].x, (int)G.bh[0].y, (int)G.bh[0].z, (int)G.bh[0].w,                                    \
                          (int)G.bh[1].x, (int)G.bh[1].y, (int)G.bh[1].z, (int)G.bh[1].w};                                   \
        const i32x8 bl = {(int)G.bl[0].x, (int)G.bl[0].y, (int)G.bl[0].z, (int)G.bl[0].w,                                    \
                          (int)G.bl[1].x, (int)G.bl[1].y, (int)G.bl[1].z, (int)G.bl[1].w};                                   \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                                                   \
            const i32x8 a = {(int)G.a[rt].x, (int)G.a[rt].y, (int)G.a[rt].z, (int)G.a[rt].w, 0, 0, 0, 0};                    \
            acc[rt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bh, acc[rt], 4, 0, S, (int)sa[rt], S, (int)sxh);    \
            acc[rt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bl, acc[rt], 4, 0, S, (int)sa[rt], S, (int)sxl);    \
        }                                                                                                                    \
    }
    // one k-step of operand loads in flight ahead of the MFMAs.  Deeper rings were tried: three register sets let the compiler
    // hoist every load of a group (450 registers, or 306 spilled under a 256 bound), four cost the second wave per SIMD (312
    // registers, 422 vs 460 TF at 1024 x 12288 x 2048).  The kernel is bound by L2 round trips, not by the matrix pipe; the next
    // step is staging the activation tile through LDS by DMA, as the MoE kernels do.
    const int nq = K >> 8;
    StepRegs<RT> g0, g1;
    unsigned sa[RT], sxh, sxl;
    load_scales(0, sa, sxh, sxl);
    load_step(0, g0);
    for (int q = 0; q < nq; ++q) {
        const int t = 4 * q;
        load_step(t + 1, g1);
        SGLK_FP4_STEP(0, g0)
        load_step(t + 2, g0);
        SGLK_FP4_STEP(1, g1)
        load_step(t + 3, g1);
        SGLK_FP4_STEP(2, g0)
        unsigned na[RT], nxh = 0, nxl = 0;
        const bool more = q + 1 < nq;
        if (more) {
            load_step(t + 4, g0);
            load_scales(q + 1, na, nxh, nxl);
        }
        SGLK_FP4_STEP(3, g1)
        if (more) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) sa[rt] = na[rt];
            sxh = nxh;
            sxl = nxl;
        }
    }
#undef SGLK_FP4_STEP
    // accumulator register i of lane (c = lane & 31, hh = lane >> 5): weight row (i & 3) + 8 (i >> 2) + 4 hh of the tile, token c
    if (tok >= p.M) return;
    uint16_t* orow = p.out + (int64_t)tok * p.out_stride;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + rt * 32 + 8 * g + 4 * h;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[rt][4 * g + j] + (p.bias ? p.bias[n + j] : 0.f);
            uint2 w;
            w.x = pack_bf16x2(v[0], v[1]);
            w.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(orow + n) = w;
        }
}

}  // namespace gfp4

size_t mxfp4_native_workspace_bytes(int M, int N, int K) {
    (void)N;
    return align_up((size_t)M * 2 * K, 256) + align_up((size_t)M * (K >> 7), 256);
}

bool mxfp4_native_ok(int M, int N, int K, const void* x, int64_t x_stride, const void* wq, const void* out, int64_t out_stride) {
    // enough 128 x 128 tiles to fill at least half the chip (smaller problems: the generic engine's split-K fills it better);
    // SGLK_MXFP4_NATIVE=1 / 0 forces the choice for every legal shape (tests, A/B)
    const int force = knobs().mxfp4_native;
    const bool big = (int64_t)(N / 128) * ceil_div(M, 128) * 2 >= device_cu_count();
    if (force == 0 || (force < 0 && !big)) return false;
    return M >= 64 && N % 128 == 0 && K % 256 == 0 && x_stride % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)wq % 16) == 0 &&
           out_stride % 4 == 0 && ((uintptr_t)out % 8) == 0 && !knobs().force_generic;
}

int launch_gemm_mxfp4_native(const void* x, int64_t x_stride, const void* wq, const void* scales, int scale_packed, const float* bias,
                             void* out, int64_t out_stride, int M, int N, int K, void* workspace, hipStream_t stream) {
    using namespace gfp4;
    uint8_t* xq = (uint8_t*)workspace;
    uint8_t* xs = xq + align_up((size_t)M * 2 * K, 256);
    hipLaunchKernelGGL(split_natural_kernel, dim3((unsigned)ceil_div(M, 4)), dim3(256), 0, stream, (const uint16_t*)x, x_stride, xq,
                       (int64_t)2 * K, xs, (int64_t)(K >> 7), (int64_t)M, K);
    Fp4Params p{};
    p.xq = xq; p.xq_stride = (int64_t)2 * K; p.xs = xs; p.xs_stride = K >> 7;
    p.wq = (const uint8_t*)wq; p.ws = (const uint8_t*)scales; p.scale_packed = scale_packed; p.bias = bias;
    p.out = (uint16_t*)out; p.out_stride = out_stride; p.M = M; p.N = N; p.K = K;
    if (knobs().mxfp4_rt == 2)
        hipLaunchKernelGGL(gemm_mxfp4_kernel<2>, dim3((unsigned)(N / 64), (unsigned)ceil_div(M, 128)), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL(gemm_mxfp4_kernel<4>, dim3((unsigned)(N / 128), (unsigned)ceil_div(M, 128)), dim3(256), 0, stream, p);
    SGLK_CHECK_LAUNCH("mxfp4_scaled_mm(native)");
    return SGLK_OK;
}

}  // namespace sglk
