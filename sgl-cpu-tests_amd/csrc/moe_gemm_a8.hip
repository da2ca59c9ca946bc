// Opt-in "a8" mode of fp8 fused_experts (sglk.h: SGLK_MOE_FP8_ACT): fp8 activations x fp8 weights on the BLOCK-SCALED fp8
// matrix cores (v_mfma_scale_f32_32x32x64_f8f6f4: one instruction = 32x32 outputs over 64 k at twice the bf16 rate per
// clock).  NOT the reference's W8A16 numerics (/root/reference/bench_moe.py:113-130 keeps bf16 activations); never a
// default.  Oracle of THIS arithmetic: oracle/moe_a8.py (quantises exactly as below, then exact sums).
//
//   * activations are quantised per token x 128-wide K block to e4m3 with a POWER-OF-TWO scale (an E8M0 byte; the smallest
//     2^e with amax / 2^e <= 448): quant_fp8_block128_kernel for `hidden`, the GATE_UP epilogue for ic1 (its 128 output
//     columns per workgroup ARE one K block of GEMM-2, so ic1 never exists in bf16: half the bytes written and re-read);
//   * the E8M0 byte goes into the MFMA as the per-lane B scale; the weights' fp32 block scale s = m * 2^e gives the A scale
//     byte (its exponent field) and the mantissa m is carried by the accumulator-unit trick of moe_gemm_fp8w_256i.hip
//     (T <- T * m_prev / m_cur per K block, C = m_last * T): fp8 x fp8 products and the power-of-two scales are exact;
//   * the quantised rows are stored in the k ORDER OF THE PACKED WEIGHT TILE (pack.hip): inside every 64-wide k group, byte
//     position p = 32 h + q holds k = 16 h + {q | 32 + q - 8 | 8 + q - 16 | 40 + q - 24} (q in 0..7 | 8..15 | 16..23 |
//     24..31), so that a lane's 32 operand bytes are two 16-byte slots of the weight tile as it lies in LDS and 32
//     contiguous bytes of the token row -- the dot product does not care about the order as long as both sides agree;
//   * tile 256 tokens x 256 weight rows, 8 waves (4 along weight rows x 2 along tokens; 64 x 128 per wave, weights = A
//     operand), K in 64-deep stages (= ONE MFMA k-step) through a ring of FOUR 32-KiB LDS buffers filled by LDS-DMA three
//     stages ahead; one counted vmcnt + one barrier per stage; the operand fragments of stage t+1 are read during the MFMAs
//     of stage t.
#include "fp8_split.h"
#include "knobs.h"
#include "moe_internal.h"

namespace sglk {

typedef __attribute__((address_space(3))) void* lptr_a8_t;
typedef __attribute__((ext_vector_type(8))) int i32x8;

namespace ga8 {

constexpr int kBM = 256;
constexpr int kStageX = kBM * 64;             // 16 KiB: 256 tokens x 64 k (fp8)
constexpr int kStageW = 16 * 1024;            // 16 KiB: 16 packed 16x64 fp8 tiles
constexpr int kStage = kStageX + kStageW;     // 32 KiB
constexpr int kRing = 4;
constexpr int kMaxKB = 32;                    // reduction length <= 4096
constexpr int kScaleOff = kRing * kStage;                 // 128 KiB: sc[16 pieces][kMaxKB] f32 (2 KiB)
constexpr int kXsOff = kScaleOff + 16 * kMaxKB * 4;       // xs[kb][256 tokens] E8M0 bytes (8 KiB)
constexpr int kRowTabOff = kXsOff + kMaxKB * kBM;         // DOWN: output slot + routing weight per tile row (2 KiB)
constexpr int kLds = kRowTabOff + 2 * kBM * 4;            // 140 KiB
constexpr int kAmaxOff = 64 * 1024;                       // GATE_UP epilogue (ring dead): amax[4 wn][256] f32

SGLK_DEV float uniform_f32(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// weight block scale s = mant * 2^(eb - 127): eb = the E8M0 byte for the MFMA, mant in +-[1,2).  Zero / denormal scales:
// eb = 0 (2^-127: the block contributes < 1e-30 instead of exactly 0), mant = 1; inf / nan: eb = 127, mant = s (poisons)
SGLK_DEV void split_scale(float s_in, int& eb, float& mant) {
    const float s = uniform_f32(s_in);
    const unsigned u = __float_as_uint(s);
    const unsigned ex = (u >> 23) & 0xffu;
    const bool tiny = ex == 0u, special = ex == 0xffu;
    eb = tiny ? 0 : (special ? 127 : (int)ex);
    mant = tiny ? 1.f : (special ? s : __uint_as_float((u & 0x807fffffu) | 0x3f800000u));
}

}  // namespace ga8

// ------------------------------------------------------------------------------------------------------------------------
// hidden [rows][cols] bf16 -> q [rows][cols] e4m3 (k order of the packed weight tile inside every 64 group) + one E8M0 byte
// per 128-wide block.  One wave per row (fp8_split.h: quant_row_block128).
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quant_fp8_block128_kernel(const uint16_t* __restrict__ x, int64_t x_stride,
                                                                 uint8_t* __restrict__ q, int64_t q_stride,
                                                                 uint8_t* __restrict__ s, int64_t s_stride, int64_t rows,
                                                                 int cols) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    quant_row_block128(x + row * x_stride, q + row * q_stride, s + row * s_stride, cols, threadIdx.x & 63);
}

int launch_quant_fp8_block128(const uint16_t* x, int64_t x_stride, uint8_t* q, int64_t q_stride, uint8_t* s, int64_t s_stride,
                              int64_t rows, int cols, hipStream_t stream) {
    if (rows == 0) return SGLK_OK;
    if (cols % 128 != 0 || x_stride % 8 != 0 || ((uintptr_t)x % 16) != 0 || q_stride % 16 != 0 || ((uintptr_t)q % 16) != 0)
        SGLK_FAIL(SGLK_ERR_SHAPE, "quant_fp8_block128: %d columns / strides / alignment not supported", cols);
    hipLaunchKernelGGL(quant_fp8_block128_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, stream, x, x_stride, q, q_stride,
                       s, s_stride, rows, cols);
    SGLK_CHECK_LAUNCH("quant_fp8_block128");
    return SGLK_OK;
}

namespace ga8 {

template <int MODE>
__global__ __launch_bounds__(512, 2) void moe_gemm_a8_kernel(const A8GemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[kLds];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wm = wave >> 2;

    // ---- tile: XCD x owns the contiguous range [xs, xs + xl) of (m-tile, column tile) pairs, column tiles fastest (the
    //      workgroups of an m-tile are neighbours and share its gathered rows; one expert's weights stay in one L2) ----
    const int nmt = p.num_tiles[0];
    const int live = nmt * p.n_tiles;
    int L;
    {
        const int x = blockIdx.x & 7, q = live >> 3, r = live & 7;
        const int xs = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        const int xl = q + (x < r ? 1 : 0);
        const int jt = blockIdx.x >> 3;
        if (jt >= xl) return;
        L = xs + jt;
    }
#ifdef SGLK_DEV_ABLATE
    const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
#define SGLK_STAMP(i) do { if (p.dbg && tid == 0) p.dbg[32 * L + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SGLK_STAMP(i) do { } while (0)
#endif
    const int mtile = L / p.n_tiles, ntile = L - mtile * p.n_tiles;
    const int4 ti = p.tile_info[mtile];
    const int e = __builtin_amdgcn_readfirstlane(ti.x);
    const int pos0 = __builtin_amdgcn_readfirstlane(ti.y);
    const int rows = __builtin_amdgcn_readfirstlane(ti.z);

    const int ctiles = p.C >> 6;      // 64-wide k groups = stages
    const int kblocks = p.C >> 7;
    const int T = ctiles;

    auto piece_row16 = [&](int piece) {
        if (MODE == MODE_GATE_UP) return (piece < 8) ? ntile * 8 + piece : (p.n_half >> 4) + ntile * 8 + (piece - 8);
        return ntile * 16 + piece;
    };

    // ---- prologue loads (parked in registers; written to the LDS tables after the first DMA stages have been issued) ----
    float* sc = reinterpret_cast<float*>(smem + kScaleOff);          // sc[piece][kb]
    unsigned char* xs_tab = smem + kXsOff;                             // xs_tab[kb][token row]
    int* slot_tab = reinterpret_cast<int*>(smem + kRowTabOff);
    float* tw_tab = reinterpret_cast<float*>(smem + kRowTabOff + kBM * 4);
    float sc_reg = 0.f;
    {
        const int piece = tid >> 5, kb = tid & (kMaxKB - 1);
        if (kb < kblocks) {
            const float* scale_e = p.w_scale + (int64_t)e * p.scale_rows * p.scale_cols;
            const int srow = (int)(((float)(piece_row16(piece) * 16) + 0.5f) * (1.0f / (float)p.block_n));
            sc_reg = scale_e[srow * p.scale_cols + kb];
        }
    }
    // the row this thread describes (tid < 256): its quantised-row index, its scale bytes, DOWN: output slot + weight
    int my_slot = -1;
    float my_tw = 0.f;
    unsigned xs_reg[kMaxKB / 4];
#pragma unroll
    for (int i = 0; i < kMaxKB / 4; ++i) xs_reg[i] = 0x7f7f7f7fu;
    if (tid < kBM && tid < rows) {
        const int slot = p.sorted_slot[pos0 + tid];
        const int64_t xrow = (MODE == MODE_GATE_UP) ? (int64_t)(slot / p.topk) : (int64_t)(pos0 + tid);
        const unsigned* sp = reinterpret_cast<const unsigned*>(p.xs + xrow * p.xs_stride);
#pragma unroll
        for (int i = 0; i < kMaxKB / 4; ++i)
            if (i * 4 < kblocks) xs_reg[i] = sp[i];
        // DOWN: the routing weight (a load that DEPENDS on `slot`) is only needed by the epilogue and is fetched near the end
        // of the main loop: here it would make the in-order wave wait for `slot` before any DMA stage could go out
        if (MODE == MODE_DOWN) my_slot = slot;
    }

    // ---- LDS-DMA sources: descriptors in SGPRs + one 32-bit lane offset per piece; the stage offset is the scalar soffset.
    //      X piece = 16 rows x 64 B (lane = row l >> 2, 16-byte chunk l & 3); image chunk = logical chunk ^ ((row >> 2) & 3),
    //      applied to the SOURCE address (the LDS destination of a DMA is lane-linear) ----
    const unsigned xbytes = (unsigned)__builtin_amdgcn_readfirstlane((int)p.x_bytes);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, xbytes, 0x00020000);
    const unsigned char* wexp = p.w + (int64_t)e * p.w_expert_stride;
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)wexp, 0, (unsigned)p.w_expert_stride, 0x00020000);
    unsigned xsrc[2], wsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 16 + (lane >> 2);
        unsigned off = xbytes;   // rows past the tile's last: out of the descriptor's range, fetches nothing
        if (r < rows) {
            int64_t xrow;
            if (MODE == MODE_GATE_UP) xrow = (int64_t)(p.sorted_slot[pos0 + r] / p.topk);
            else xrow = (int64_t)(pos0 + r);
            off = (unsigned)(xrow * p.x_stride) + (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
        }
        xsrc[i] = off;
        wsrc[i] = (unsigned)(piece_row16(wave * 2 + i) * ctiles) * 1024u + lane * 16;
    }
#ifdef SGLK_A8_ABLATE   // developer A/B builds only (-DSGLK_A8_ABLATE=bits; wrong results by design): 1 no rescale, 2 no DMA
    constexpr int abl = SGLK_A8_ABLATE;   // after the prologue, 4 no fragment reads in the loop
#else
    constexpr int abl = 0;
#endif
    auto issue_piece = [&](int kt, int buf, int i) {   // i = 0,1: X pieces; 2,3: W pieces of this wave
        if ((abl & 2) && kt > 3) return;
        unsigned char* sx = smem + buf * kStage;
        if (i < 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lptr_a8_t)(sx + (wave * 2 + i) * 1024), 16, xsrc[i], kt * 64, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lptr_a8_t)(sx + kStageX + (wave * 2 + i - 2) * 1024), 16,
                                                     wsrc[i - 2], kt * 1024, 0, 0);
    };

    // ---- operand addressing (lane l: r32 = l & 31 = operand row / column, h = l >> 5 = which 32 of the stage's 64 k) ----
    // A (weights), row tile rt: 16-row piece wpiece0[rt] + (r32 >> 4); the lane's 32 bytes = slots g = 2h, 2h + 1 of row
    // r32 & 15 in the piece's lane-linear image: byte offsets ((2h) * 16 + (r32 & 15)) * 16 and + 256
    const int h = lane >> 5, r32 = lane & 31;
    int wpiece0[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        if (MODE == MODE_GATE_UP) wpiece0[rt] = rt == 0 ? wn * 2 : 8 + wn * 2;    // gate rows, matching up rows
        else wpiece0[rt] = wn * 4 + rt * 2;
    }
    // row tile 1 sits a constant number of pieces behind row tile 0 (8 gate pieces / 2), so ONE lane offset serves both
    // (the rest is the instruction's immediate offset)
    constexpr int kRt1 = (MODE == MODE_GATE_UP ? 8 : 2) * 1024;
    const int woff0 = kStageX + (wpiece0[0] + (r32 >> 4)) * 1024 + ((2 * h) * 16 + (r32 & 15)) * 16;
    // B (tokens), token tile tt: row = wm * 128 + tt * 32 + r32; chunks 2h, 2h + 1, each ^ ((row >> 2) & 3).  The swizzle
    // term only depends on r32 (the tile bases are multiples of 16 rows), so token tile tt is + tt * 2048 bytes: immediate
    const int row0 = wm * 128 + r32;
    const int xoff0 = row0 * 64 + (((2 * h) ^ ((r32 >> 2) & 3)) << 4);
    const int xoff1 = row0 * 64 + (((2 * h + 1) ^ ((r32 >> 2) & 3)) << 4);

    f32x16 acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][tt][i] = 0.f;

    // ---- prologue: all four ring slots in flight, tables to LDS, stage 0 landed ----
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_piece(st, st, i);
    sc[tid] = sc_reg;
    if (tid < kBM) {
#pragma unroll
        for (int i = 0; i < kMaxKB / 4; ++i)
            if (i * 4 < kblocks) {
#pragma unroll
                for (int b = 0; b < 4; ++b) xs_tab[(i * 4 + b) * kBM + tid] = (unsigned char)(xs_reg[i] >> (8 * b));
            }
        if (MODE == MODE_DOWN) slot_tab[tid] = my_slot;
    }
    asm volatile("s_waitcnt vmcnt(12)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    int ea[2], ea_next[2];
    float mant[2], ratio[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        split_scale(sc[wpiece0[rt] * kMaxKB], ea[rt], mant[rt]);
        ea_next[rt] = ea[rt];
        ratio[rt] = 1.f;
    }
    // B scale bytes of the lane's four tokens for the current / next K block
    int xsv[4], xsv_next[4];
    auto ld_xs = [&](int kb, int* dst) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) dst[tt] = xs_tab[kb * kBM + wm * 128 + tt * 32 + r32];
    };
    ld_xs(0, xsv);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) xsv_next[tt] = xsv[tt];

#define SGLK_FENCE() __builtin_amdgcn_sched_barrier(0)
    i32x8 fa[2] = {}, fb[4] = {};
    float nsc[2] = {0.f, 0.f};
    auto ld_a = [&](i32x8& dst, int rt, int buf) {
        if (abl & 4) return;
        const unsigned char* b = smem + (buf * kStage + woff0);
        const i32x4 lo = *reinterpret_cast<const i32x4*>(b + rt * kRt1);
        const i32x4 hi = *reinterpret_cast<const i32x4*>(b + rt * kRt1 + 256);
        dst = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto ld_b = [&](i32x8& dst, int tt, int buf) {
        if (abl & 4) return;
        const i32x4 lo = *reinterpret_cast<const i32x4*>(smem + (buf * kStage + xoff0) + tt * 2048);
        const i32x4 hi = *reinterpret_cast<const i32x4*>(smem + (buf * kStage + xoff1) + tt * 2048);
        dst = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    // MFMA slot s of a stage: row tile s >> 2, token tile s & 3.  Row tile 0's fragment is free after slot 3 and token
    // fragment tt after slot 4 + tt, so the next stage's fragments are read INTO THE SAME REGISTERS behind their last use (no
    // second register set: 128 accumulators + 48 operand registers); the two that free up last (row tile 1, token tile 3)
    // are read at the start of their own stage and first used in its slots 4 / 3.
    auto mma = [&](int s2) {
        const int rt = s2 >> 2, tt = s2 & 3;
        acc[rt][tt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[rt], fb[tt], acc[rt][tt], 0, 0, 0, ea[rt], 0, xsv[tt]);
    };
    auto rescale = [&](int s2) {   // accumulator of slot s2 into units of the next K block's mantissa
        const int rt = s2 >> 2, tt = s2 & 3;
        if (abl & 1) return;
#pragma unroll
        for (int i = 0; i < 16; ++i) asm("v_mul_f32 %0, %1, %0" : "+v"(acc[rt][tt][i]) : "s"(ratio[rt]));
    };
    auto sync_point = [&](int wait) {
        if (wait == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (wait == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    const bool active = wm * 128 < rows;
    int buf = 0;
#ifdef SGLK_DEV_ABLATE
    if (p.dbg && tid == 0) {
        p.dbg[32 * L + 18] = rt_entry;
        p.dbg[32 * L + 22] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
        p.dbg[32 * L + 23] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
    }
#endif
    SGLK_STAMP(19);
    // Stage t (ring slot `buf`); every flag is a literal at the call site.
    //   first   : first stage of a K block -- the scales switch; with resc_hi accumulators 6, 7 are rescaled (slots 0, 1)
    //   bound   : closing stage of a K block that is not the last -- rescale accumulators 0..5 in slots 2..7
    //   wait    : >= 0: stage t+1 exists; sync point after slot 1 = `s_waitcnt vmcnt(wait)` (this wave's pieces of stage t+1
    //             have landed, later ones stay in flight) + lgkmcnt(0) + barrier; the fragments of stage t+1 are read after it
    //   dma     : stage t+4 exists: its four pieces go into THIS stage's ring slot, which nobody reads after the sync point
    //   own     : read this stage's row tile 1 / token tile 3 in slot 0 (false only for stage 0, preloaded)
    //   pre     : first stage of a K block that is not the last: the NEXT block's scales (weight scale of both row tiles, the
    //             four token scale bytes) are read from the LDS tables here, a whole stage before `bound` consumes them
    auto stage = [&](int t, bool first, bool bound, int wait, bool dma, bool resc_hi, bool own, bool pre) {
        int nbuf = buf + 1;
        if (nbuf == kRing) nbuf = 0;
        if (first) {   // this block's scales (read one block ago) become current ...
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) ea[rt] = ea_next[rt];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) xsv[tt] = xsv_next[tt];
        }
        if (pre) {     // ... BEFORE the next block's are requested into the same registers
            const int kb = (t >> 1) + 1;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) nsc[rt] = sc[wpiece0[rt] * kMaxKB + kb];
            ld_xs(kb, xsv_next);
        }
        if (bound) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                float nm;
                split_scale(nsc[rt], ea_next[rt], nm);
                ratio[rt] = uniform_f32(mant[rt] * __builtin_amdgcn_rcpf(nm));
                mant[rt] = nm;
            }
        }
        mma(0);
        SGLK_FENCE();
        if (own) { ld_b(fb[3], 3, buf); ld_a(fa[1], 1, buf); }
        if (first && resc_hi) rescale(6);
        SGLK_FENCE();
        mma(1);
        SGLK_FENCE();
        if (first && resc_hi) rescale(7);
        if (wait >= 0) sync_point(wait);
        SGLK_FENCE();
        mma(2);
        SGLK_FENCE();
        if (dma) { issue_piece(t + 4, buf, 0); issue_piece(t + 4, buf, 1); }
        if (bound) rescale(0);
        SGLK_FENCE();
        mma(3);
        SGLK_FENCE();
        if (dma) { issue_piece(t + 4, buf, 2); issue_piece(t + 4, buf, 3); }
        if (bound) rescale(1);
        SGLK_FENCE();
        mma(4);
        SGLK_FENCE();
        if (wait >= 0) ld_a(fa[0], 0, nbuf);
        if (bound) rescale(2);
        SGLK_FENCE();
        mma(5);
        SGLK_FENCE();
        if (wait >= 0) ld_b(fb[0], 0, nbuf);
        if (bound) rescale(3);
        SGLK_FENCE();
        mma(6);
        SGLK_FENCE();
        if (wait >= 0) ld_b(fb[1], 1, nbuf);
        if (bound) rescale(4);
        SGLK_FENCE();
        mma(7);
        SGLK_FENCE();
        if (wait >= 0) ld_b(fb[2], 2, nbuf);
        if (bound) rescale(5);
        SGLK_FENCE();
        buf = nbuf;
    };
    // waves without rows (tail tiles): keep the DMA pieces and the sync points, skip the math
    auto idle_stage = [&](int t, int wait, bool dma) {
        if (wait >= 0) sync_point(wait);
        if (dma) {
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_piece(t + 4, buf, i);
        }
        buf = (buf + 1 == kRing) ? 0 : buf + 1;
    };

    // T = 2 * kblocks >= 4.  Steady state while stage t+4 exists for both stages of the K block; the last two K blocks are
    // peeled so that the wait counts stay literals.
    if (active) {
        ld_a(fa[0], 0, 0);
        ld_a(fa[1], 1, 0);
        ld_b(fb[0], 0, 0);
        ld_b(fb[1], 1, 0);
        ld_b(fb[2], 2, 0);
        ld_b(fb[3], 3, 0);
        SGLK_FENCE();
        int t = 0;
        if (kblocks > 2) {
            stage(0, true, false, 8, true, false, false, true);
            stage(1, false, true, 8, true, false, true, false);
            for (t = 2; t + 4 < T; t += 2) {
                stage(t, true, false, 8, true, true, true, true);
                stage(t + 1, false, true, 8, true, true, true, false);
            }
            // t == T - 4
            stage(t, true, false, 8, false, true, true, true);
            stage(t + 1, false, true, 4, false, true, true, false);
            if (MODE == MODE_DOWN && my_slot >= 0) my_tw = p.topk_weights[my_slot];   // covered by the next stage's vmcnt(0)
            stage(t + 2, true, false, 0, false, true, true, false);
            stage(t + 3, false, false, -1, false, true, true, false);
        } else {   // kblocks == 2: stages 0..3, no stage 4
            stage(0, true, false, 8, false, false, false, true);
            stage(1, false, true, 4, false, false, true, false);
            if (MODE == MODE_DOWN && my_slot >= 0) my_tw = p.topk_weights[my_slot];
            stage(2, true, false, 0, false, true, true, false);
            stage(3, false, false, -1, false, true, true, false);
        }
    } else {
        int t = 0;
        for (; t + 4 < T; ++t) idle_stage(t, 8, true);
        idle_stage(t, 8, false);
        idle_stage(t + 1, 4, false);
        idle_stage(t + 2, 0, false);
        idle_stage(t + 3, -1, false);
    }
#undef SGLK_FENCE
    SGLK_STAMP(20);
    if (MODE == MODE_DOWN && tid < kBM) tw_tab[tid] = my_tw;   // rows of tile rows 0..255 = waves 0..3, always active

    // ---- epilogue (ring dead).  32x32 accumulator: lane = token column (l & 31); register i = weight row
    //      (i & 3) + 8 * (i >> 2) + 4 * (l >> 5) of the row tile ----
    __syncthreads();
    SGLK_STAMP(25);
    int tidv = tid;
    asm volatile("" : "+v"(tidv));
    const int r32e = tidv & 31, he = (tidv >> 5) & 1;
    if (MODE == MODE_GATE_UP) {
        // ic1 = silu(gate) * up for this workgroup's 128 columns = ONE K block of GEMM-2: per-token amax over the four waves
        // along n, power-of-two scale, e4m3, stored in the packed-tile k order (see the file header)
        float* amax_tab = reinterpret_cast<float*>(smem + kAmaxOff);
        float v[4][16];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            float am = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float g = acc[0][tt][i] * mant[0], u = acc[1][tt][i] * mant[1];
                v[tt][i] = active ? silu_f32(g) * u : 0.f;
                am = fmaxf(am, fabsf(v[tt][i]));
            }
            am = fmaxf(am, __shfl_xor(am, 32));
            if (he == 0) amax_tab[wn * kBM + wm * 128 + tt * 32 + r32e] = am;
        }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int r = wm * 128 + tt * 32 + r32e;
            const float am = fmaxf(fmaxf(amax_tab[r], amax_tab[kBM + r]), fmaxf(amax_tab[2 * kBM + r], amax_tab[3 * kBM + r]));
            const int sb = e8m0_for_amax(am);
            const float inv = inv_scale_of(sb);
            if (wn == 0 && he == 0 && r < rows) p.out_s[(int64_t)(pos0 + r) * p.out_s_stride + ntile] = (uint8_t)sb;
            unsigned char* rowp = smem + r * 128;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                int d = 0;
                d = __builtin_amdgcn_cvt_pk_fp8_f32(v[tt][rg * 4 + 0] * inv, v[tt][rg * 4 + 1] * inv, d, false);
                d = __builtin_amdgcn_cvt_pk_fp8_f32(v[tt][rg * 4 + 2] * inv, v[tt][rg * 4 + 3] * inv, d, true);
                // columns wn*32 + rg*8 + he*4 .. +3 of the 128: 64 group wn >> 1, k = (wn & 1)*32 + rg*8 + he*4
                //   -> position 32*(rg >> 1) + 8*(wn & 1) + 16*(rg & 1) + 4*he
                const int pos = (wn >> 1) * 64 + 32 * (rg >> 1) + 8 * (wn & 1) + 16 * (rg & 1) + 4 * he;
                const int chunk = (pos >> 4) ^ (r & 7);
                *reinterpret_cast<int*>(rowp + chunk * 16 + (pos & 15)) = d;
            }
        }
        SGLK_STAMP(26);
        __syncthreads();
        SGLK_STAMP(27);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 512 + tidv;
            const int r = idx >> 3, pc = idx & 7, lc = pc ^ (r & 7);
            if (r < rows) {
                const uint4 val = *reinterpret_cast<const uint4*>(smem + r * 128 + pc * 16);
                *reinterpret_cast<uint4*>((unsigned char*)p.out + (int64_t)(pos0 + r) * p.out_stride + ntile * 128 + lc * 16) = val;
            }
        }
    } else {
        // ic2[slot] = topk_w * (acc * mant) in bf16: XOR-swizzled [token][256 columns] image, whole rows out by slot
        constexpr int kRowB = 512;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            if (!active) break;
            const int r = wm * 128 + tt * 32 + r32e;
            unsigned char* rowp = smem + r * kRowB;
            const float tw = tw_tab[r];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const float sc_w = mant[rt] * tw;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    uint2 val;
                    val.x = pack_bf16x2(acc[rt][tt][rg * 4 + 0] * sc_w, acc[rt][tt][rg * 4 + 1] * sc_w);
                    val.y = pack_bf16x2(acc[rt][tt][rg * 4 + 2] * sc_w, acc[rt][tt][rg * 4 + 3] * sc_w);
                    const int col = wn * 64 + rt * 32 + rg * 8 + he * 4;
                    const int chunk = (col >> 3) ^ (r & 15);
                    *reinterpret_cast<uint2*>(rowp + chunk * 16 + (col & 4) * 2) = val;
                }
            }
        }
        SGLK_STAMP(26);
        __syncthreads();
        SGLK_STAMP(27);
        uint16_t* outp = reinterpret_cast<uint16_t*>(p.out);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = it * 512 + tidv;
            const int r = idx >> 5, pc = idx & 31, lc = pc ^ (r & 15);
            if (r < rows) {
                const uint4 val = *reinterpret_cast<const uint4*>(smem + r * kRowB + pc * 16);
                *reinterpret_cast<uint4*>(outp + (int64_t)slot_tab[r] * p.out_stride + ntile * 256 + lc * 8) = val;
            }
        }
    }
#ifdef SGLK_DEV_ABLATE
    if (p.dbg && tid == 0) {
        p.dbg[32 * L + 21] = __builtin_amdgcn_s_memrealtime();   // stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p.dbg[32 * L + 24] = __builtin_amdgcn_s_memrealtime();   // stores acknowledged
    }
#endif
}

}  // namespace ga8

int launch_moe_gemm_a8(int mode, const A8GemmParams& p, int max_mtiles, hipStream_t stream) {
    const int64_t blocks = (int64_t)max_mtiles * p.n_tiles;
    if (blocks == 0) return SGLK_OK;
    const int kblocks = p.C >> 7;
    if (p.C % 128 != 0 || kblocks < 2 || kblocks > ga8::kMaxKB)
        SGLK_FAIL(SGLK_ERR_SHAPE, "moe_gemm_a8: reduction length %d (needs 2..%d whole 128-wide K blocks)", p.C, ga8::kMaxKB);
    if (p.block_n % 32 != 0) SGLK_FAIL(SGLK_ERR_SHAPE, "moe_gemm_a8: block_n %d is not a multiple of 32", p.block_n);
    if (p.xs_stride % 4 != 0 || ((uintptr_t)p.xs % 4) != 0) SGLK_FAIL(SGLK_ERR_INVALID, "moe_gemm_a8: scale rows must be 4-byte aligned");
    if (mode == MODE_GATE_UP)
        hipLaunchKernelGGL((ga8::moe_gemm_a8_kernel<MODE_GATE_UP>), dim3((unsigned)blocks), dim3(512), 0, stream, p);
    else if (mode == MODE_DOWN)
        hipLaunchKernelGGL((ga8::moe_gemm_a8_kernel<MODE_DOWN>), dim3((unsigned)blocks), dim3(512), 0, stream, p);
    else
        SGLK_FAIL(SGLK_ERR_INVALID, "moe_gemm_a8: mode %d", mode);
    SGLK_CHECK_LAUNCH("moe_gemm_a8");
    return SGLK_OK;
}

}  // namespace sglk
