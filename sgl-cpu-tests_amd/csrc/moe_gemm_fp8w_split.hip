// Large-M grouped W8A16 GEMM of fused_experts on the BLOCK-SCALED fp8 matrix cores, with bf16 activations kept EXACT by a
// two-term split (same operator contract and oracle as moe_gemm_fp8w_256i.hip: /root/reference/test_moe_fp8_ext.py:22-25,70-91;
// bench_moe.py:113-130).
//
// Why: v_mfma_scale_f32_32x32x64_f8f6f4 multiplies fp8 x fp8 at twice the bf16 rate and takes the weights AS THEY ARE (no
// fp8 -> bf16 conversion, which costs the bf16 kernel 8 quarter-rate VALU instructions per k-step and half of its MFMA issue
// slots).  A bf16 activation has 8 significant bits, an e4m3 value 4: x = hi + lo with
//     hi = e4m3(x / s),   lo = e4m3((x - hi * s) / (s / 16)),   s = 2^sb the block's power-of-two scale (amax / s <= 448)
// is EXACT for every element within 2^13 of its block's largest magnitude (hi is then a normal e4m3 number) (the residual of the first rounding has at most
// four significant bits and sits at most 2^-4 below the element); smaller elements lose bits below 2^-21 * amax(block) --
// four orders of magnitude under the bf16 rounding of the result.  Two scaled MFMAs (hi, lo) per 64-wide k group cost the
// matrix pipe exactly what the four bf16 MFMAs of the same k range cost; products stay exact, accumulation fp32, the block
// scales exact (power of two in the instruction, mantissa by the accumulator-unit trick), ic1 rounded to bf16 once and then
// split the same way -- the numerics contract of the bf16 kernel, which the same parity tests hold it to.
//
// Data: a split row stores, for every 64-wide k group, [hi 64 B | lo 64 B] in the k order of the packed weight tile (see
// moe_gemm_a8.hip) = 128 contiguous bytes per token and stage (one cache line); one E8M0 byte per token and 128-wide block.
// Tile 256 tokens x 256 weight rows, 8 waves (4 along weight rows x 2 along tokens, 64 x 128 per wave), K in 64-deep stages
// through a ring of THREE 48-KiB LDS buffers (X 32 KiB + W 16 KiB) filled by LDS-DMA; 16 MFMAs per wave and stage.
#include "knobs.h"
#include "moe_internal.h"
#include "fp8_split.h"

namespace sglk {

typedef __attribute__((address_space(3))) void* lptr_sp_t;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// ------------------------------------------------------------------------------------------------------------------------
// hidden [rows][cols] bf16 -> q [rows][2 * cols] ([hi 64 | lo 64] per 64 group, packed-tile k order) + one E8M0 byte per
// 128-wide block.  One wave per row (fp8_split.h: split_row_block128).
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_fp8_block128_kernel(const uint16_t* __restrict__ x, int64_t x_stride,
                                                                 uint8_t* __restrict__ q, int64_t q_stride,
                                                                 uint8_t* __restrict__ s, int64_t s_stride, int64_t rows,
                                                                 int cols) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    split_row_block128(x + row * x_stride, q + row * q_stride, s + row * s_stride, cols, threadIdx.x & 63);
}

int launch_split_fp8_block128(const uint16_t* x, int64_t x_stride, uint8_t* q, int64_t q_stride, uint8_t* s, int64_t s_stride,
                              int64_t rows, int cols, hipStream_t stream) {
    if (rows == 0) return SGLK_OK;
    if (cols % 128 != 0 || x_stride % 8 != 0 || ((uintptr_t)x % 16) != 0 || q_stride % 16 != 0 || ((uintptr_t)q % 16) != 0)
        SGLK_FAIL(SGLK_ERR_SHAPE, "split_fp8_block128: %d columns / strides / alignment not supported", cols);
    hipLaunchKernelGGL(split_fp8_block128_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, stream, x, x_stride, q, q_stride,
                       s, s_stride, rows, cols);
    SGLK_CHECK_LAUNCH("split_fp8_block128");
    return SGLK_OK;
}

namespace gsp {

constexpr int kBM = 256;
constexpr int kStageX = kBM * 128;            // 32 KiB: 256 tokens x (hi 64 + lo 64) bytes
constexpr int kStageW = 16 * 1024;            // 16 KiB: 16 packed 16x64 fp8 tiles
constexpr int kStage = kStageX + kStageW;     // 48 KiB
constexpr int kRing = 3;
constexpr int kMaxKB = 32;                    // reduction length <= 4096
constexpr int kScaleOff = kRing * kStage;                 // 144 KiB: sc[16 pieces][kMaxKB] f32 (2 KiB)
constexpr int kXsOff = kScaleOff + 16 * kMaxKB * 4;       // xs[kb][256 tokens] E8M0 bytes (8 KiB)
constexpr int kRowTabOff = kXsOff + kMaxKB * kBM;         // DOWN: output slot + routing weight per tile row (2 KiB)
constexpr int kLds = kRowTabOff + 2 * kBM * 4;            // 156 KiB
constexpr int kAmaxOff = 96 * 1024;                       // GATE_UP epilogue (ring dead): amax[4 wn][256] f32, above the 64-KiB image

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

template <int MODE>
__global__ __launch_bounds__(512, 2) void moe_gemm_fp8w_split_kernel(const A8GemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[kLds];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wm = wave >> 2;

    // ---- tile: XCD x owns the contiguous range [xs, xs + xl) of (m-tile, column tile) pairs, column tiles fastest ----
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

    auto piece_row16 = [&](int piece) __attribute__((always_inline)) {
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
        // DOWN: the routing weight (a load that DEPENDS on `slot`) is only needed by the epilogue and is fetched near the end of
        // the main loop: here it would make the in-order wave wait for `slot` before any DMA stage could go out
        if (MODE == MODE_DOWN) my_slot = slot;
    }

    // ---- LDS-DMA sources: descriptors in SGPRs + one 32-bit lane offset per piece; the stage offset is the scalar soffset.
    //      X piece = 8 rows x 128 B (lane = row l >> 3, 16-byte chunk l & 7); image chunk = logical chunk ^ ((row >> 1) & 7),
    //      applied to the SOURCE address (the LDS destination of a DMA is lane-linear) ----
    const unsigned xbytes = (unsigned)__builtin_amdgcn_readfirstlane((int)p.x_bytes);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, xbytes, 0x00020000);
    const unsigned char* wexp = p.w + (int64_t)e * p.w_expert_stride;
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)wexp, 0, (unsigned)p.w_expert_stride, 0x00020000);
    unsigned xsrc[4], wsrc[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        unsigned off = xbytes;   // rows past the tile's last: out of the descriptor's range, fetches nothing
        if (r < rows) {
            int64_t xrow;
            if (MODE == MODE_GATE_UP) xrow = (int64_t)(p.sorted_slot[pos0 + r] / p.topk);
            else xrow = (int64_t)(pos0 + r);
            off = (unsigned)(xrow * p.x_stride) + (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) << 4);
        }
        xsrc[i] = off;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) wsrc[i] = (unsigned)(piece_row16(wave * 2 + i) * ctiles) * 1024u + lane * 16;
    auto issue_piece = [&](int kt, int buf, int i) __attribute__((always_inline)) {   // i = 0..3: X pieces; 4,5: W pieces of this wave
        unsigned char* sx = smem + buf * kStage;
        if (i < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lptr_sp_t)(sx + (wave * 4 + i) * 1024), 16, xsrc[i], kt * 128, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lptr_sp_t)(sx + kStageX + (wave * 2 + i - 4) * 1024), 16,
                                                     wsrc[i - 4], kt * 1024, 0, 0);
    };

    // ---- operand addressing (lane l: r32 = l & 31 = operand row / column, h = l >> 5 = which 32 of the stage's 64 k) ----
    const int h = lane >> 5, r32 = lane & 31;
    int wpiece0[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        if (MODE == MODE_GATE_UP) wpiece0[rt] = rt == 0 ? wn * 2 : 8 + wn * 2;    // gate rows, matching up rows
        else wpiece0[rt] = wn * 4 + rt * 2;
    }
    constexpr int kRt1 = (MODE == MODE_GATE_UP ? 8 : 2) * 1024;
    const int woff0 = kStageX + (wpiece0[0] + (r32 >> 4)) * 1024 + ((2 * h) * 16 + (r32 & 15)) * 16;
    // B (tokens), token tile tt: row = wm * 128 + tt * 32 + r32; hi chunks 2h, 2h + 1, lo chunks 4 + 2h, 5 + 2h, each
    // ^ ((row >> 1) & 7), which only depends on r32; token tile tt is + tt * 4096 bytes (an immediate)
    const int row0 = wm * 128 + r32;
    const int sw = (r32 >> 1) & 7;
    const int xo_h0 = row0 * 128 + (((2 * h) ^ sw) << 4), xo_h1 = row0 * 128 + (((2 * h + 1) ^ sw) << 4);
    const int xo_l0 = row0 * 128 + (((4 + 2 * h) ^ sw) << 4), xo_l1 = row0 * 128 + (((5 + 2 * h) ^ sw) << 4);

    f32x16 acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][tt][i] = 0.f;

    // ---- prologue: all three ring slots in flight, tables to LDS, stage 0 landed ----
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
        for (int i = 0; i < 6; ++i) issue_piece(st, st, i);
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
    // B scale bytes of the lane's four tokens for the current / next K block (the lo term's scale is this - 4)
    int xsv[4], xsl[4], xsv_next[4];
    auto ld_xs = [&](int kb, int* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) dst[tt] = xs_tab[kb * kBM + wm * 128 + tt * 32 + r32];
    };
    ld_xs(0, xsv);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) { xsv_next[tt] = xsv[tt]; xsl[tt] = xsv[tt] - 4; }

#define SGLK_FENCE() __builtin_amdgcn_sched_barrier(0)
    i32x8 fa[2][2] = {};                   // [stage parity][row tile]: the next stage's weights are read beside the current one's
    i32x8 bh[2] = {}, bl[2] = {};          // token-tile window: set tt & 1 holds the hi / lo fragments of token tile tt
    float nsc[2] = {0.f, 0.f};
    auto ld_a = [&](i32x8& dst, int rt, int buf) __attribute__((always_inline)) {
        const unsigned char* b = smem + (buf * kStage + woff0);
        const i32x4 lo = *reinterpret_cast<const i32x4*>(b + rt * kRt1);
        const i32x4 hi = *reinterpret_cast<const i32x4*>(b + rt * kRt1 + 256);
        dst = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto ld_bh = [&](int tt, int buf) __attribute__((always_inline)) {
        const i32x4 a0 = *reinterpret_cast<const i32x4*>(smem + (buf * kStage + xo_h0) + tt * 4096);
        const i32x4 a1 = *reinterpret_cast<const i32x4*>(smem + (buf * kStage + xo_h1) + tt * 4096);
        bh[tt & 1] = i32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    };
    auto ld_bl = [&](int tt, int buf) __attribute__((always_inline)) {
        const i32x4 a0 = *reinterpret_cast<const i32x4*>(smem + (buf * kStage + xo_l0) + tt * 4096);
        const i32x4 a1 = *reinterpret_cast<const i32x4*>(smem + (buf * kStage + xo_l1) + tt * 4096);
        bl[tt & 1] = i32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    };
    // MFMA slot s of a stage (16 per wave): token tile s >> 2, then hi x rt0, hi x rt1, lo x rt0, lo x rt1
    auto mma = [&](int par, int s2) __attribute__((always_inline)) {
        const int tt = s2 >> 2, lo = (s2 >> 1) & 1, rt = s2 & 1;
        if (lo)
            acc[rt][tt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[par][rt], bl[tt & 1], acc[rt][tt], 0, 0, 0, ea[rt], 0, xsl[tt]);
        else
            acc[rt][tt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[par][rt], bh[tt & 1], acc[rt][tt], 0, 0, 0, ea[rt], 0, xsv[tt]);
    };
    auto rescale = [&](int rt, int tt) __attribute__((always_inline)) {   // accumulator into units of the next K block's mantissa
#pragma unroll
        for (int i = 0; i < 16; ++i) asm("v_mul_f32 %0, %1, %0" : "+v"(acc[rt][tt][i]) : "s"(ratio[rt]));
    };
    auto sync_point = [&](int wait) __attribute__((always_inline)) {
        if (wait == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    const bool active = wm * 128 < rows;
    int buf = 0;
    SGLK_STAMP(19);
    // Stage t (ring slot `buf`); every flag is a literal at the call site.
    //   first   : first stage of a K block -- the scales switch; with resc_hi token tile 3's accumulators are rescaled (slots 1, 2)
    //   pre     : first stage of a K block that is not the last: the NEXT block's scales are read from the LDS tables
    //   bound   : closing stage of a K block that is not the last -- rescale token tiles 0..2 (slots 5,6 / 9,10 / 13,14)
    //   wait    : >= 0: stage t+1 exists; sync point after slot 13 = `s_waitcnt vmcnt(wait)` (this wave's pieces of stage t+1
    //             have landed, stage t+2's may stay in flight) + lgkmcnt(0) + barrier.  Every fragment of THIS stage has been
    //             read by then, so afterwards stage t+3's pieces go into this stage's ring slot and the first fragments of
    //             stage t+1 are read
    //   dma     : stage t+3 exists
    //   par     : t & 1 as a literal (which of the two weight-fragment register sets this stage multiplies with)
    auto stage = [&](int par, int t, bool first, bool bound, int wait, bool dma, bool resc_hi, bool pre, bool dma_carry) __attribute__((always_inline)) {
        int nbuf = buf + 1;
        if (nbuf == kRing) nbuf = 0;
        int pbuf = buf - 1;
        if (pbuf < 0) pbuf = kRing - 1;
        if (first) {   // this block's scales (read one block ago) become current ...
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) ea[rt] = ea_next[rt];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) { xsv[tt] = xsv_next[tt]; xsl[tt] = xsv_next[tt] - 4; }
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
        // token tile 0 (window set 0, read at the end of the previous stage); token tile 1 arrives meanwhile
        mma(par, 0);
        SGLK_FENCE();
        ld_bh(1, buf);
        SGLK_FENCE();
        mma(par, 1);
        SGLK_FENCE();
        ld_bl(1, buf);
        if (first && resc_hi) rescale(0, 3);
        SGLK_FENCE();
        mma(par, 2);
        SGLK_FENCE();
        if (dma_carry) { issue_piece(t + 2, pbuf, 4); }     // the previous stage's last two DMA pieces ride here
        if (first && resc_hi) rescale(1, 3);
        SGLK_FENCE();
        mma(par, 3);
        SGLK_FENCE();
        if (dma_carry) { issue_piece(t + 2, pbuf, 5); }
        SGLK_FENCE();
        // token tile 1; token tile 2 arrives
        mma(par, 4);
        SGLK_FENCE();
        ld_bh(2, buf);
        SGLK_FENCE();
        mma(par, 5);
        SGLK_FENCE();
        ld_bl(2, buf);
        if (bound) rescale(0, 0);
        SGLK_FENCE();
        mma(par, 6);
        SGLK_FENCE();
        if (bound) rescale(1, 0);
        SGLK_FENCE();
        mma(par, 7);
        SGLK_FENCE();
        // token tile 2; token tile 3 arrives
        mma(par, 8);
        SGLK_FENCE();
        ld_bh(3, buf);
        SGLK_FENCE();
        mma(par, 9);
        SGLK_FENCE();
        ld_bl(3, buf);
        if (bound) rescale(0, 1);
        SGLK_FENCE();
        mma(par, 10);
        SGLK_FENCE();
        if (bound) rescale(1, 1);
        SGLK_FENCE();
        mma(par, 11);
        SGLK_FENCE();
        // token tile 3: its hi MFMAs keep the pipe fed across the sync point
        mma(par, 12);
        SGLK_FENCE();
        mma(par, 13);
        SGLK_FENCE();
        if (bound) rescale(0, 2);
        if (wait >= 0) sync_point(wait);
        SGLK_FENCE();
        mma(par, 14);
        SGLK_FENCE();
        if (wait >= 0) { ld_a(fa[par ^ 1][0], 0, nbuf); ld_a(fa[par ^ 1][1], 1, nbuf); }
        if (dma) { issue_piece(t + 3, buf, 0); issue_piece(t + 3, buf, 1); }
        if (bound) rescale(1, 2);
        SGLK_FENCE();
        mma(par, 15);
        SGLK_FENCE();
        if (wait >= 0) { ld_bh(0, nbuf); ld_bl(0, nbuf); }
        if (dma) { issue_piece(t + 3, buf, 2); issue_piece(t + 3, buf, 3); }
        SGLK_FENCE();
        buf = nbuf;
    };
    // waves without rows (tail tiles): keep the DMA pieces and the sync points, skip the math
    auto idle_stage = [&](int t, int wait, bool dma, bool dma_carry) __attribute__((always_inline)) {
        int pbuf = buf - 1;
        if (pbuf < 0) pbuf = kRing - 1;
        if (dma_carry) { issue_piece(t + 2, pbuf, 4); issue_piece(t + 2, pbuf, 5); }
        if (wait >= 0) sync_point(wait);
        if (dma) {
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_piece(t + 3, buf, i);
        }
        buf = (buf + 1 == kRing) ? 0 : buf + 1;
    };

    // DMA bookkeeping: stage t issues pieces 0..3 of stage t+3 (after its sync point) and stage t+1 issues the remaining
    // pieces 4,5 of that same stage t+3 = (t+1)+2 in its slots 2,3 (`dma_carry`).  At the sync point of stage t (which asserts
    // stage t+1) the wave's younger DMA instructions are exactly the six pieces of stage t+2 -> vmcnt(6).
    // T = 2 * kblocks >= 4; the last stages are peeled so that the wait counts stay literals.
    if (active) {
        ld_a(fa[0][0], 0, 0);
        ld_a(fa[0][1], 1, 0);
        ld_bh(0, 0);
        ld_bl(0, 0);
        SGLK_FENCE();
        int t = 0;
        // stage 0: nothing carried in (the prologue issued stages 0..2 whole)
        stage(0, 0, true, false, 6, T > 3, false, kblocks > 1, false);
        stage(1, 1, false, true, 6, T > 4, false, false, T > 3);
        for (t = 2; t + 4 < T; t += 2) {
            stage(0, t, true, false, 6, true, true, true, true);
            stage(1, t + 1, false, true, 6, true, true, false, true);
        }
        // t == T - 4 or T - 2
        if (t + 3 < T) {   // stages T-4, T-3: stage T-1 is the last one DMA is issued for
            stage(0, t, true, false, 6, true, true, true, true);                // issues T-1 (pieces 0..3)
            stage(1, t + 1, false, true, 6, false, true, false, true);          // carries pieces 4,5 of T-1
            t += 2;
        }
        if (MODE == MODE_DOWN && my_slot >= 0) my_tw = p.topk_weights[my_slot];   // covered by the next sync point's vmcnt(0)
        stage(0, t, true, false, 0, false, true, false, false);                 // T-2: waits for all of T-1
        stage(1, t + 1, false, false, -1, false, true, false, false);           // T-1: nothing follows
    } else {
        int t = 0;
        idle_stage(0, 6, T > 3, false);
        idle_stage(1, 6, T > 4, T > 3);
        for (t = 2; t + 4 < T; t += 2) {
            idle_stage(t, 6, true, true);
            idle_stage(t + 1, 6, true, true);
        }
        if (t + 3 < T) {
            idle_stage(t, 6, true, true);
            idle_stage(t + 1, 6, false, true);
            t += 2;
        }
        idle_stage(t, 0, false, false);
        idle_stage(t + 1, -1, false, false);
    }
#undef SGLK_FENCE
    SGLK_STAMP(20);
    if (MODE == MODE_DOWN && tid < kBM) tw_tab[tid] = my_tw;   // tile rows 0..255 = waves 0..3, always active

    // ---- epilogue (ring dead).  32x32 accumulator: lane = token column (l & 31); register i = weight row
    //      (i & 3) + 8 * (i >> 2) + 4 * (l >> 5) of the row tile ----
    __syncthreads();
    SGLK_STAMP(25);
    int tidv = tid;
    asm volatile("" : "+v"(tidv));
    const int r32e = tidv & 31, he = (tidv >> 5) & 1;
    if (MODE == MODE_GATE_UP) {
        // ic1 = bf16(silu(gate) * up) -- rounded to bf16 ONCE, as the bf16 kernel does -- for this workgroup's 128 columns = one K
        // block of GEMM-2, then split exactly like `hidden`: per-token amax over the four waves along n, power-of-two scale,
        // (hi, lo), stored [hi 64 | lo 64] per 64 group in the packed-tile k order
        float* amax_tab = reinterpret_cast<float*>(smem + kAmaxOff);
        float v[4][16];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            float am = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const float g0 = acc[0][tt][i] * mant[0], u0 = acc[1][tt][i] * mant[1];
                const float g1 = acc[0][tt][i + 1] * mant[0], u1 = acc[1][tt][i + 1] * mant[1];
                const unsigned pk = active ? pack_bf16x2(silu_f32(g0) * u0, silu_f32(g1) * u1) : 0u;
                v[tt][i] = __uint_as_float(pk << 16);
                v[tt][i + 1] = __uint_as_float(pk & 0xffff0000u);
                am = fmaxf(am, fmaxf(fabsf(v[tt][i]), fabsf(v[tt][i + 1])));
            }
            am = fmaxf(am, __shfl_xor(am, 32));
            if (he == 0) amax_tab[wn * kBM + wm * 128 + tt * 32 + r32e] = am;
        }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int r = wm * 128 + tt * 32 + r32e;
            const float am = fmaxf(fmaxf(amax_tab[r], amax_tab[kBM + r]), fmaxf(amax_tab[2 * kBM + r], amax_tab[3 * kBM + r]));
            const int sb = sp_e8m0_for_amax(am);
            if (wn == 0 && he == 0 && r < rows) p.out_s[(int64_t)(pos0 + r) * p.out_s_stride + ntile] = (uint8_t)sb;
            unsigned char* rowp = smem + r * 256;      // image row: [group 0: hi 64 | lo 64][group 1: hi 64 | lo 64]
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {           // register groups 2rp, 2rp + 1 = eight values = two dwords of hi and of lo
                unsigned hi[2], lo[2];
                split8(&v[tt][rp * 8], sb, hi, lo);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int rg = rp * 2 + q;
                    // columns wn*32 + rg*8 + he*4 .. +3 of the 128: 64 group wn >> 1, k = (wn & 1)*32 + rg*8 + he*4
                    //   -> position 32*(rg >> 1) + 8*(wn & 1) + 16*(rg & 1) + 4*he inside the group's hi (and lo) half
                    const int pos = 32 * (rg >> 1) + 8 * (wn & 1) + 16 * (rg & 1) + 4 * he;
                    const int bh_ = (wn >> 1) * 128 + pos, bl_ = bh_ + 64;
                    *reinterpret_cast<unsigned*>(rowp + (((bh_ >> 4) ^ (r & 15)) << 4) + (bh_ & 15)) = hi[q];
                    *reinterpret_cast<unsigned*>(rowp + (((bl_ >> 4) ^ (r & 15)) << 4) + (bl_ & 15)) = lo[q];
                }
            }
        }
        SGLK_STAMP(26);
        __syncthreads();
        SGLK_STAMP(27);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 512 + tidv;
            const int r = idx >> 4, pc = idx & 15, lc = pc ^ (r & 15);
            if (r < rows) {
                const uint4 val = *reinterpret_cast<const uint4*>(smem + r * 256 + pc * 16);
                *reinterpret_cast<uint4*>((unsigned char*)p.out + (int64_t)(pos0 + r) * p.out_stride + ntile * 256 + lc * 16) = val;
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
        p.dbg[32 * L + 18] = rt_entry;
        p.dbg[32 * L + 22] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
        p.dbg[32 * L + 23] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
        p.dbg[32 * L + 21] = __builtin_amdgcn_s_memrealtime();   // stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p.dbg[32 * L + 24] = __builtin_amdgcn_s_memrealtime();   // stores acknowledged
    }
#endif
}

}  // namespace gsp

int launch_moe_gemm_fp8w_split(int mode, const A8GemmParams& p, int max_mtiles, hipStream_t stream) {
    const int64_t blocks = (int64_t)max_mtiles * p.n_tiles;
    if (blocks == 0) return SGLK_OK;
    const int kblocks = p.C >> 7;
    if (p.C % 128 != 0 || kblocks < 2 || kblocks > gsp::kMaxKB)
        SGLK_FAIL(SGLK_ERR_SHAPE, "moe_gemm_fp8w_split: reduction length %d (needs 2..%d whole 128-wide K blocks)", p.C, gsp::kMaxKB);
    if (p.block_n % 32 != 0) SGLK_FAIL(SGLK_ERR_SHAPE, "moe_gemm_fp8w_split: block_n %d is not a multiple of 32", p.block_n);
    if (p.xs_stride % 4 != 0 || ((uintptr_t)p.xs % 4) != 0) SGLK_FAIL(SGLK_ERR_INVALID, "moe_gemm_fp8w_split: scale rows must be 4-byte aligned");
    if (mode == MODE_GATE_UP)
        hipLaunchKernelGGL((gsp::moe_gemm_fp8w_split_kernel<MODE_GATE_UP>), dim3((unsigned)blocks), dim3(512), 0, stream, p);
    else if (mode == MODE_DOWN)
        hipLaunchKernelGGL((gsp::moe_gemm_fp8w_split_kernel<MODE_DOWN>), dim3((unsigned)blocks), dim3(512), 0, stream, p);
    else
        SGLK_FAIL(SGLK_ERR_INVALID, "moe_gemm_fp8w_split: mode %d", mode);
    SGLK_CHECK_LAUNCH("moe_gemm_fp8w_split");
    return SGLK_OK;
}

}  // namespace sglk
