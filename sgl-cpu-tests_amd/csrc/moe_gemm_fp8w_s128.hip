// Large-M grouped W8A16 GEMM of fused_experts, two-term e4m3 split on the block-scaled fp8 matrix cores, 128-token tiles,
// TWO workgroups per CU.  Operator contract, oracle and data formats: fp8_split.h
// (/root/reference/test_moe_fp8_ext.py:22-25,70-91; /root/reference/bench_moe.py:113-130): x = hi + lo exactly, both terms e4m3
// under one power-of-two scale per (row, 128-wide block); rows are [hi 64 | lo 64] per 64-wide k group in the packed weight
// tile's k order.
//
// Why this tiling.  A 256 x 256 / 8-wave form of the same split (round 2; deleted in round 3) ran its main loop at 91 % of the
// matrix pipe but a tile also spent 3.6 us in its prologue (tile table -> row ids -> first operands: dependent round trips) and
// 7.2 us in its epilogue (SiLU, the split of ic1, stores), with the pipe idle: one workgroup owned the whole CU.  Here a workgroup is FOUR waves (one
// per SIMD) with a tile of 128 tokens x 256 weight rows, and a CU holds TWO of them (2 x 71 KiB of LDS, 2 x 256 registers
// per SIMD lane): while one is in its prologue or epilogue the other has the matrix pipe to itself, and a 64-cycle
// v_mfma_scale_f32_32x32x64_f8f6f4 leaves one wave alone enough issue slots to keep the pipe full.
//   * waves 4 (weight rows) x 1: a wave owns 64 weight rows (GATE_UP: 32 gate + the 32 matching up rows) for ALL 128 tokens
//     = 2 x 4 accumulator tiles of 32 x 32 (128 registers), 16 MFMAs per 64-deep stage (hi and lo term);
//   * weights never touch LDS: no two waves share a weight row, so a wave's A fragments of a stage are four
//     buffer_load_dwordx4 straight from the packed tiles (pack.hip; a lane's 2 x 16 bytes are its operand bytes), issued two
//     stages ahead into a rotation of three register sets;
//   * activations go through a ring of four 16-KiB LDS buffers filled by LDS-DMA (rows gathered through sorted_slot by the
//     per-lane source address), three stages ahead; one counted s_waitcnt + one barrier per stage;
//   * half the tile height of the 256-row kernels, and an expert's last tile skips the 32-token tiles that hold no row.
// Not built in: a persistent form (two resident workgroups per CU, the next tile's row ids and first stages fetched inside the
// epilogue) was written and measured in round 3 (git 9d5f90f): with the tile loop around it the function sits at the limit of
// both register files, ROCm 7.2's allocator spills ~170 values per tile, and every scratch reload waits behind the loads in
// flight -- 0.90 vs 0.65 ms for GEMM-1.
#include <stddef.h>

#include <type_traits>

#include "fp8_split.h"
#include "knobs.h"
#include "moe_internal.h"

namespace sglk {

typedef __attribute__((address_space(3))) void* lptr_s1_t;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) int i32x16;

namespace gs128 {

constexpr int kBM = 128;
#ifndef SGLK_S128_GROUP
#define SGLK_S128_GROUP 8
#endif
constexpr int kTileGroup = SGLK_S128_GROUP;   // consecutive m-tiles dealt to one XCD (A/B: 4 and 16 measured the same or worse)
constexpr int kStageX = kBM * 128;            // 16 KiB: 128 tokens x (hi 64 + lo 64) bytes
constexpr int kRing = 4;
constexpr int kMaxKB = 64;                    // reduction length <= 8192
constexpr int kImage = 64 * 1024;             // the ring (4 x 16 KiB) = the DOWN epilogue image (128 rows x 512 B)
constexpr int kScaleOff = kImage;                         // sc[16 pieces][kMaxKB] f32 (4 KiB)
constexpr int kXsOff = kScaleOff + 16 * kMaxKB * 4;       // xs[kb][128 tokens] E8M0 bytes (8 KiB)
constexpr int kRowTabOff = kXsOff + kMaxKB * kBM;         // DOWN: output slot + routing weight per tile row (1 KiB)
constexpr int kAmaxOff = kRowTabOff + 2 * kBM * 4;        // GATE_UP epilogue: amax[4 waves][128] f32 (2 KiB)
constexpr int kLds = kAmaxOff + 4 * kBM * 4;              // 79 KiB: two workgroups per CU
constexpr int kScPer = 16 * kMaxKB / 256;                 // scale-table entries a thread fetches

// s_waitcnt immediate (gfx9 encoding): vmcnt in bits 3:0 and 15:14, expcnt 6:4 (7 = no wait), lgkmcnt 11:8 (15 = no wait)
constexpr int wc(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

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

enum { KB_STEADY = 0, KB_PENULT = 1, KB_LAST = 2 };

struct TileId {
    int L, mt, ntile, e, pos0, rows;   // L = linear tile id (time stamps of developer builds only), mt = m-tile
};

// One tile.
// NMOD = (kblocks - 2) % 3: stage t multiplies with weight-fragment set (t + const) % 3, and the rotation is laid out from the
// END of the reduction so that every set index is a literal (a run-time phase switch around the K-block bodies cost the
// register allocator 1400 spills): K blocks in phases P0 (sets 0,1), P1 (2,0), P2 (1,2); the last two blocks are always P1, P2,
// the steady blocks before them whole (P1, P2, P0) triples preceded by a head of NMOD blocks (P0, or P2 P0).  (Four sets --
// weights three stages ahead -- measured the same and cost 16 registers.)
// NTA = token tiles (32 tokens) of the tile that hold at least one row: 4 for full tiles, 1..3 for an expert's last tile, which
// skips the fragment reads, MFMAs and rescales of the token tiles without rows.  The whole tile is compiled per count (a
// run-time test in front of every MFMA, or a switch around the main loop alone, cost the register allocator 30-100 spills).
// TERMS = e4m3 terms per activation: 2 = the exact two-term split of the bf16 value (W8A16, the reference's numerics); 1 = the
// opt-in a8 mode (activations QUANTISED to e4m3 per token x 128 block: fp8_split.h: quant_row_block128, oracle/moe_a8.py; half the MFMAs,
// X bytes and LDS reads per stage).
// I8 = the int8 W8A8 operator (/root/reference/test_moe_int8.py:59-94, bench_moe.py:89-106) on the same pipeline: one int8 term
// per activation (TERMS = 1 data movement), weights in pack.hip's int8 tiles (natural k order), two mfma_i32_32x32x32_i8 per
// 64-deep slot (the lane's two 16-byte halves; exact int32 sums), NO scales inside the loop: the per-token and per-weight-row
// factors are applied in the epilogue in the oracle's order.  GATE_UP quantises silu(gate) * up per token over the WHOLE row
// (the oracle's per_token_quant_int8), which spans the m-tile's n_tiles workgroups: they exchange the row maxima through
// global atomics and a per-m-tile arrival counter (see the epilogue).
template <int MODE, int NMOD, int ABL, int NTA, int TERMS, bool I8>
SGLK_DEV void run_tile(const A8GemmParams& p, unsigned char* smem, const TileId cur) {
    static_assert(!I8 || TERMS == 1, "int8 moves one byte per activation");
    constexpr int kXB = 64 * TERMS;             // activation bytes per token and stage
    constexpr int kStageXT = kBM * kXB;         // ring slot: 16 KiB (two terms) / 8 KiB
    constexpr int kXP = 2 * TERMS;              // 1-KiB LDS-DMA pieces of a stage per wave
    constexpr int kSP = 8 * TERMS;              // MFMAs of a stage per wave
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ctiles = p.C >> 6;      // 64-wide k groups = stages
    const int kblocks = p.C >> 7;
#ifdef SGLK_DEV_ABLATE
    const int L = cur.L;
    const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
#define SGLK_STAMP(i) do { if (p.dbg && tid == 0) p.dbg[32 * L + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SGLK_STAMP(i) do { } while (0)
#endif

    if (p.prio == 2) __builtin_amdgcn_s_setprio(1);
    float* sc = reinterpret_cast<float*>(smem + kScaleOff);          // sc[piece][kb]
    unsigned char* xs_tab = smem + kXsOff;                             // xs_tab[kb][token row]
    int* slot_tab = reinterpret_cast<int*>(smem + kRowTabOff);
    float* tw_tab = reinterpret_cast<float*>(smem + kRowTabOff + kBM * 4);
    float* amax_tab = reinterpret_cast<float*>(smem + kAmaxOff);

    // the workgroup's 16 packed 16-row weight pieces: GATE_UP = 8 gate + the 8 matching up pieces, DOWN = 16 consecutive
    auto piece_row16 = [&](int piece) __attribute__((always_inline)) {
        if (MODE == MODE_GATE_UP) return (piece < 8) ? cur.ntile * 8 + piece : (p.n_half >> 4) + cur.ntile * 8 + (piece - 8);
        return cur.ntile * 16 + piece;
    };
    // W: lane l (r32 = l & 31 = operand row, h = l >> 5 = which 32 of the stage's 64 k) of 32-row tile rt takes slots
    // (2h) * 16 + (r32 & 15) and + 16 of packed piece wpiece0[rt] + (r32 >> 4): 2 x 16 bytes = the operand's 32 bytes
    const int h = lane >> 5, r32 = lane & 31;
    int wpiece0[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        if (MODE == MODE_GATE_UP) wpiece0[rt] = rt == 0 ? wn * 2 : 8 + wn * 2;    // gate rows, matching up rows
        else wpiece0[rt] = wn * 4 + rt * 2;
    }
    // B (tokens), token tile tt: row = tt * 32 + r32; hi chunks 2h, 2h + 1, lo chunks 4 + 2h, 5 + 2h, each ^ ((row >> 1) & 7),
    // which only depends on r32; token tile tt is + tt * 4096 bytes (an immediate)
    // (one term: rows of 64 bytes, chunks 2h, 2h + 1 ^ ((row >> 2) & 3), token tile tt + tt * 2048)
    const int sw = TERMS == 2 ? (r32 >> 1) & 7 : (r32 >> 2) & 3;
    const int xo_h0 = r32 * kXB + (((2 * h) ^ sw) << 4), xo_h1 = r32 * kXB + (((2 * h + 1) ^ sw) << 4);
    const int xo_l0 = r32 * kXB + (((4 + 2 * h) ^ sw) << 4), xo_l1 = r32 * kXB + (((5 + 2 * h) ^ sw) << 4);   // two terms only

    // ---- prologue loads (parked in registers; written to the LDS tables after the first operand loads have been issued) ----
    float sc_reg[kScPer] = {};
    if (I8) {   // per-weight-row factors of the workgroup's 256 rows: sc[piece * 16 + row of the piece]
        sc_reg[0] = p.w_scale[(int64_t)cur.e * p.scale_rows + piece_row16(tid >> 4) * 16 + (tid & 15)];
    } else {
        const float* scale_e = p.w_scale + (int64_t)cur.e * p.scale_rows * p.scale_cols;
        const float inv_bn = 1.0f / (float)p.block_n;
#pragma unroll
        for (int j = 0; j < kScPer; ++j) {
            const int i = tid + j * 256, piece = i / kMaxKB, kb = i & (kMaxKB - 1);
            sc_reg[j] = 0.f;
            if (kb < kblocks) {
                // floor(row / block_n) through one float multiply (exact for rows < 2^20, see moe_gemm_fp8w_256i.hip)
                const int srow = (int)(((float)(piece_row16(piece) * 16) + 0.5f) * inv_bn);
                sc_reg[j] = scale_e[srow * p.scale_cols + kb];
            }
        }
    }
    float bias_reg = 0.f;   // PLAIN: the output column's bias -> the (unused) row table's place
    if (MODE == MODE_PLAIN && p.bias) bias_reg = p.bias[cur.ntile * 256 + tid];
    int my_slot = -1;
    unsigned xs_reg[kMaxKB / 4];
    float xs_f32 = 0.f;   // int8: the token's dequantisation factor
#pragma unroll
    for (int i = 0; i < kMaxKB / 4; ++i) xs_reg[i] = 0x7f7f7f7fu;
    if (tid < kBM && tid < cur.rows) {
        const int slot = MODE == MODE_PLAIN ? 0 : p.sorted_slot[cur.pos0 + tid];   // PLAIN (dense rows): no routing tables
        const int64_t xrow = (MODE == MODE_GATE_UP) ? (int64_t)(slot / p.topk) : (int64_t)(cur.pos0 + tid);
        if (I8) {
            xs_f32 = p.x_scale_f32[xrow];
        } else {
            const unsigned* sp = reinterpret_cast<const unsigned*>(p.xs + xrow * p.xs_stride);
#pragma unroll
            for (int i = 0; i < kMaxKB / 4; ++i)
                if (i * 4 < kblocks) xs_reg[i] = sp[i];
        }
        if (MODE == MODE_DOWN) my_slot = slot;   // its routing weight (a dependent load) is fetched near the end of the main loop
    }

    // ---- operand sources.  X: descriptor + one 32-bit lane offset per 1-KiB piece (8 rows x 128 B: lane = row l >> 3, chunk
    //      l & 7; image chunk = logical chunk ^ ((row >> 1) & 7), applied to the SOURCE address since an LDS-DMA lands
    //      lane-linear); the stage offset is the scalar soffset.  Rows past the tile's last one get an offset outside the
    //      descriptor's range and fetch nothing.  W: descriptor of the expert + one lane offset per 32-row tile. ----
    const unsigned xbytes = (unsigned)__builtin_amdgcn_readfirstlane((int)p.x_bytes);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (int64_t)cur.e * p.w_expert_stride), 0,
                                                                           (unsigned)p.w_expert_stride, 0x00020000);
    unsigned xsrc[kXP], wsrc[2];
#pragma unroll
    for (int i = 0; i < kXP; ++i) {
        // a piece = 1 KiB = 8 rows x 128 B (two terms) or 16 rows x 64 B
        const int r = TERMS == 2 ? (wn * 4 + i) * 8 + (lane >> 3) : (wn * 2 + i) * 16 + (lane >> 2);
        unsigned off = xbytes;
        if (r < cur.rows) {
            int64_t xrow;
            if (MODE == MODE_GATE_UP) xrow = (int64_t)(p.sorted_slot[cur.pos0 + r] / p.topk);
            else xrow = (int64_t)(cur.pos0 + r);
            const unsigned chunk = TERMS == 2 ? (unsigned)((lane & 7) ^ ((r >> 1) & 7)) : (unsigned)((lane & 3) ^ ((r >> 2) & 3));
            off = (unsigned)(xrow * p.x_stride) + (chunk << 4);
        }
        xsrc[i] = off;
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
        wsrc[rt] = (unsigned)(piece_row16(wpiece0[rt] + (r32 >> 4)) * ctiles) * 1024u + (unsigned)(((2 * h) * 16 + (r32 & 15)) * 16);
    auto issue_x = [&](int kt, int buf, int i) __attribute__((always_inline)) {   // piece i (0..3) of this wave, stage kt
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lptr_s1_t)(smem + buf * kStageXT + (wn * kXP + i) * 1024), 16, xsrc[i], kt * kXB, 0, 0);
    };
    i32x8 fa[TERMS == 1 ? 4 : 3][2] = {};   // [(stage + const) % 3][row tile]; one term: [stage % 4][row tile], weights three stages ahead
    auto ld_a = [&](int as, int rt, int kt) __attribute__((always_inline)) {
        const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wsrc[rt], kt * 1024, 0);
        const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wsrc[rt], kt * 1024 + 256, 0);   // + 16 slots: in the scalar offset
        fa[as][rt] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    };

    auto ld_a_half = [&](int as, int rt, int kt, int half) __attribute__((always_inline)) {   // one of the two loads of ld_a
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wsrc[rt], kt * 1024 + half * 256, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[as][rt][half * 4 + i] = (int)v[i];
    };

    // ---- prologue: X(0) A(0) X(1) A(1) X(2) and half of X(3) in flight (in this order: the counted waits rely on it) ----
    // One term (a stage is half as long in time, so everything is requested further ahead: the ring holds EIGHT 8-KiB stages,
    // the weights come three stages ahead into four register sets): X(0..4) A(0) X(5) A(1) X(6) A(2) X(7) -- the order the steady
    // state would have issued them in.  Stages past the end of a short reduction are requested all the same (they land in ring
    // slots nobody reads; the counted waits stay the same for every length).
    constexpr int kSet0 = TERMS == 1 ? 2 * NMOD : (NMOD == 0 ? 2 : (NMOD == 1 ? 0 : 1));   // fragment set of stage 0 = first set of K block 0's phase
    if constexpr (TERMS == 1) {
#pragma unroll
        for (int st = 0; st < 5; ++st) {
            issue_x(st, st, 0);
            issue_x(st, st, 1);
        }
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            ld_a((kSet0 + st) % 4, 0, st);
            ld_a((kSet0 + st) % 4, 1, st);
            issue_x(5 + st, 5 + st, 0);
            issue_x(5 + st, 5 + st, 1);
        }
    } else {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int i = 0; i < kXP; ++i) issue_x(st, st, i);
            ld_a((kSet0 + st) % 3, 0, st);
            ld_a((kSet0 + st) % 3, 1, st);
        }
#pragma unroll
        for (int i = 0; i < kXP; ++i) issue_x(2, 2, i);
        issue_x(3, 3, 0);   // two terms: pieces 2, 3 of X(3) are carried into stage 0 like every later stage's
        issue_x(3, 3, 1);
    }
    sc[tid] = sc_reg[0];
    if (!I8) {
#pragma unroll
        for (int j = 1; j < kScPer; ++j) sc[tid + j * 256] = sc_reg[j];
    }
    if (tid < kBM) {
        if (I8) {
            reinterpret_cast<float*>(xs_tab)[tid] = xs_f32;
        } else {
#pragma unroll
            for (int i = 0; i < kMaxKB / 4; ++i)
                if (i * 4 < kblocks) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) xs_tab[(i * 4 + b) * kBM + tid] = (unsigned char)(xs_reg[i] >> (8 * b));
                }
        }
        if (MODE == MODE_DOWN) slot_tab[tid] = my_slot;
    }
    if (MODE == MODE_PLAIN) reinterpret_cast<float*>(smem + kRowTabOff)[tid] = bias_reg;
    // X(0) and A(0) have landed; X(1) A(1) X(2) + two pieces of X(3) = 14 operations stay in flight
    // (one term: X(5) A(1) X(6) A(2) X(7) = 14 as well)
    __builtin_amdgcn_s_waitcnt(wc(14, 0));
    __builtin_amdgcn_s_barrier();

    typedef typename std::conditional<I8, i32x16, f32x16>::type acc_t;   // int8: exact int32 sums
    acc_t acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][tt][i] = 0;
    float my_tw = 0.f;

    int ea[2] = {0, 0}, ea_next[2] = {0, 0};
    float mant[2] = {1.f, 1.f}, ratio[2] = {1.f, 1.f};
    if (!I8) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            split_scale(sc[wpiece0[rt] * kMaxKB], ea[rt], mant[rt]);
            ea_next[rt] = ea[rt];
            ratio[rt] = 1.f;
        }
    }
    // B scale bytes of the lane's four tokens for the current K block (the lo term's scale is this - 4); token tile tt's byte
    // of the NEXT block is read into the same register right behind the tile's last MFMA of the block
    int xsv[4] = {0, 0, 0, 0};
    if (!I8) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) xsv[tt] = xs_tab[tt * 32 + r32];
    }

#define SGLK_FENCE() __builtin_amdgcn_sched_barrier(0)
    // ONE window of token fragments (hi / lo of a 32-token tile): the next tile's fragment is requested right behind the last
    // MFMA that reads the current one, two to three MFMAs (128+ cycles) ahead of its first use.  (A second window, requests
    // four MFMAs ahead, measured the same: 0.6637 vs 0.6600 ms.)
    // ABL (developer builds, wrong results by design): 1 = no accumulator rescale, 2 = no activation DMA in the steady state,
    // 4 = no weight loads in the steady state
    // One term: two windows of hi fragments, set tt & 1 for token tile tt, the next tile's requested behind the first of the
    // current tile's two MFMAs.
    i32x8 bh = {}, bl = {};     // two terms: hi / lo window; one term: windows 0 / 1
    float nsc[2] = {0.f, 0.f};
    auto ld_bh = [&](int tt, int buf) __attribute__((always_inline)) {
        const i32x4 a0 = *reinterpret_cast<const i32x4*>(smem + (buf * kStageXT + xo_h0) + tt * (32 * kXB));
        const i32x4 a1 = *reinterpret_cast<const i32x4*>(smem + (buf * kStageXT + xo_h1) + tt * (32 * kXB));
        const i32x8 f = i32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        if (TERMS == 2 || (tt & 1) == 0) bh = f;
        else bl = f;
    };
    auto ld_bl = [&](int tt, int buf) __attribute__((always_inline)) {   // two terms only
        const i32x4 a0 = *reinterpret_cast<const i32x4*>(smem + (buf * kStageXT + xo_l0) + tt * 4096);
        const i32x4 a1 = *reinterpret_cast<const i32x4*>(smem + (buf * kStageXT + xo_l1) + tt * 4096);
        bl = i32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    };
    // MFMA slot s of a stage.  Two terms (16 per wave): token tile s >> 2, then hi x rt0, hi x rt1, lo x rt0, lo x rt1; one term
    // (8): token tile s >> 1, then rt0, rt1
    auto mma = [&](int as, int s2) __attribute__((always_inline)) {
        if constexpr (TERMS == 2) {
            const int tt = s2 >> 2, lo = (s2 >> 1) & 1, rt = s2 & 1;
            if (lo)
                acc[rt][tt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[as][rt], bl, acc[rt][tt], 0, 0, 0, ea[rt], 0, xsv[tt] - 4);
            else
                acc[rt][tt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[as][rt], bh, acc[rt][tt], 0, 0, 0, ea[rt], 0, xsv[tt]);
        } else if constexpr (I8) {
            const int tt = s2 >> 1, rt = s2 & 1;
            const i32x8 a = fa[as][rt], b = (tt & 1) ? bl : bh;   // the same k runs in both halves of A and B
            acc[rt][tt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(i32x4{a[0], a[1], a[2], a[3]}, i32x4{b[0], b[1], b[2], b[3]}, acc[rt][tt], 0, 0, 0);
            acc[rt][tt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(i32x4{a[4], a[5], a[6], a[7]}, i32x4{b[4], b[5], b[6], b[7]}, acc[rt][tt], 0, 0, 0);
        } else {
            const int tt = s2 >> 1, rt = s2 & 1;
            acc[rt][tt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[as][rt], (tt & 1) ? bl : bh, acc[rt][tt], 0, 0, 0, ea[rt], 0, xsv[tt]);
        }
    };
    // accumulator tile (rt, tt) into units of the next K block's mantissa, four registers (chunk c) at a time: a wave that has
    // its SIMD to itself is issue-bound, so the 128 multiplies of a K-block boundary are spread four per tile and MFMA slot
    auto rescale4 = [&](int rt, int tt, int c) __attribute__((always_inline)) {
        if (SGLK_ABL(ABL, 1) || I8) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) asm("v_mul_f32 %0, %1, %0" : "+v"(acc[rt][tt][c * 4 + i]) : "s"(ratio[rt]));
    };
    // chunks due behind MFMA slot g (0 .. 2 kSP - 1 over the closing stage and the first stage of the next block): tile (rt, tt)
    // is free behind its last MFMA of the block (two terms: slot 4 tt + 2 + rt; one: 2 tt + rt) until its first MFMA of the
    // next block; chunk c goes behind the slot c places after that last MFMA
    auto rescale_slot = [&](int g) __attribute__((always_inline)) {
#pragma unroll
        for (int tt = 0; tt < NTA; ++tt)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int c = g - ((TERMS == 2 ? 4 * tt + 2 : 2 * tt) + rt + 1);
                if (c >= 0 && c < 4) rescale4(rt, tt, c);
            }
    };

    int buf = 0;
    SGLK_STAMP(19);
    if (p.prio == 1) __builtin_amdgcn_s_setprio(1);        // the main loop outranks a co-resident workgroup's prologue /
    else if (p.prio == 2) __builtin_amdgcn_s_setprio(0);   // epilogue (1), or the other way round (2: A/B only)
#ifdef SGLK_DEV_ABLATE
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
#endif
    // Stage t (ring slot `buf` = t % 4, weight fragments fa[as]); every flag is a literal at the call site.
    //   first  : first stage of a K block -- the weight scale switches, the rest of the boundary's rescale chunks
    //   pre    : first stage of a K block that is not the last: the NEXT block's weight scales are read from the LDS table
    //   bound  : closing stage of a K block that is not the last -- rescale chunks, and the next block's activation scale bytes
    //            are read (behind each token tile's last MFMA)
    //   lda    : stage t+2 exists: its weight fragments are requested (slots 0, 4) into fa[(as + 2) % 3], dead since stage t-1
    //   wait   : >= 0: stage t+1 exists; sync point = s_waitcnt vmcnt(N) (X(t+1) and A(t+1) have landed; the operations issued
    //            behind them -- two terms: the last piece of X(t+2), half of X(t+3), A(t+2), the rest of X(t+3) -- may stay in
    //            flight; N = 9 / 5 / 0 (two terms), 6 / 4 / 0 (one) for wait = 2 (steady) / 1 (stage T-3) / 0 (stage T-2), from
    //            replaying the issue order: tools/s128_waits.py) + lgkmcnt(0) + barrier.  Every fragment of THIS stage has been read by then, so afterwards X(t+4) goes into this stage's
    //            ring slot and the first fragments of stage t+1 are read
    //   dmax   : stage t+4 exists: pieces 0, 1 of X(t+4) go out behind slots 14, 15 (one LDS-DMA per slot: a piece costs the
    //            issuing wave 60-180 cycles, more than one MFMA's shadow), pieces 2, 3 are carried into stage t+1
    //   carry  : pieces 2, 3 of X(t+3) (slots 2, 6), whose pieces 0, 1 the previous stage issued
    auto stage = [&](int as, int t, bool first, bool bound, int wait, bool dmax, bool lda, bool pre, bool carry) __attribute__((always_inline)) {
        int nbuf = buf + 1;
        if (nbuf == kRing) nbuf = 0;
        const int as2 = (as + 2) % 3;
        if (first && !I8) {   // this block's weight scale (computed one stage ago) becomes current
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) ea[rt] = ea_next[rt];
        }
        if (pre && !I8) {
            const int kb = (t >> 1) + 1;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) nsc[rt] = sc[wpiece0[rt] * kMaxKB + kb];
        }
        if (bound && !I8) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                float nm;
                split_scale(nsc[rt], ea_next[rt], nm);
                ratio[rt] = uniform_f32(mant[rt] * __builtin_amdgcn_rcpf(nm));
                mant[rt] = nm;
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < kSP; ++s2) {
            const int tt = TERMS == 2 ? s2 >> 2 : s2 >> 1, q = TERMS == 2 ? s2 & 3 : s2 & 1;
            if (tt < NTA) mma(as, s2);
            SGLK_FENCE();
            if (TERMS == 2) {   // the hi fragment is free after the tile's second MFMA, the lo fragment after its fourth
                if (q == 1 && tt + 1 < NTA) ld_bh(tt + 1, buf);
                if (q == 3 && tt + 1 < NTA) ld_bl(tt + 1, buf);
            } else {            // the other window has been free since the previous tile's second MFMA
                if (q == 0 && tt + 1 < NTA) ld_bh(tt + 1, buf);
            }
            if (q == kSP / 4 - 1 && bound && tt < NTA && !I8) xsv[tt] = xs_tab[((t >> 1) + 1) * kBM + tt * 32 + r32];
            if (s2 == 0 && lda && !SGLK_ABL(ABL, 4)) ld_a(as2, 0, t + 2);
            if (s2 == kSP / 4 && lda && !SGLK_ABL(ABL, 4)) ld_a(as2, 1, t + 2);
            if (TERMS == 2 && s2 == 2 && carry && !SGLK_ABL(ABL, 2)) issue_x(t + 3, buf == 0 ? kRing - 1 : buf - 1, 2);
            if (TERMS == 2 && s2 == 6 && carry && !SGLK_ABL(ABL, 2)) issue_x(t + 3, buf == 0 ? kRing - 1 : buf - 1, 3);
            if (bound && !I8) rescale_slot(s2);
            if (first && !I8) rescale_slot(kSP + s2);
            if (s2 == kSP - 3 + (TERMS == 1) && wait >= 0) {   // behind slot 13 (two terms) / 6: every fragment of this stage has been read
                if (wait == 2) __builtin_amdgcn_s_waitcnt(wc(TERMS == 2 ? 9 : 6, 0));
                else if (wait == 1) __builtin_amdgcn_s_waitcnt(wc(TERMS == 2 ? 5 : 4, 0));
                else __builtin_amdgcn_s_waitcnt(wc(0, 0));
                __builtin_amdgcn_s_barrier();
                ld_bh(0, nbuf);
            }
            if (s2 == kSP - 2 && dmax && !SGLK_ABL(ABL, 2)) issue_x(t + 4, buf, 0);
            if (TERMS == 2 && s2 == 15 && wait >= 0) ld_bl(0, nbuf);
            if (s2 == kSP - 1 && dmax && !SGLK_ABL(ABL, 2)) issue_x(t + 4, buf, 1);
            SGLK_FENCE();
        }
        buf = nbuf;
    };
    // One 128-wide K block = two stages in phase ph (a literal: sets (2 ph) % 3 and (2 ph + 1) % 3); kind (a literal):
    //   STEADY : both stages request their t+2 weights and t+4 activations
    //   PENULT : block kblocks-2: no activations are left to request (X(T) does not exist), stage T-3 still requests A(T-1)
    //   LAST   : block kblocks-1: stage T-2 waits for everything, stage T-1 has nothing to wait for
    auto kblock = [&](int ph, int kind, int kb) __attribute__((always_inline)) {
        const int t = 2 * kb, a0 = (2 * ph) % 3, a1 = (2 * ph + 1) % 3;
        if (kind == KB_STEADY) {
            stage(a0, t, true, false, 2, true, true, true, true);
            stage(a1, t + 1, false, true, 2, true, true, false, true);
        } else if (kind == KB_PENULT) {
            stage(a0, t, true, false, 2, false, true, true, true);
            stage(a1, t + 1, false, true, 1, false, true, false, false);
        } else {
            if (MODE == MODE_DOWN && my_slot >= 0) my_tw = p.topk_weights[my_slot];   // covered by stage T-2's vmcnt(0)
            stage(a0, t, true, false, 0, false, false, false, false);
            stage(a1, t + 1, false, false, -1, false, false, false, false);
        }
    };
    // ---- one term: the deep form.  Stage t (ring slot t % 8, fragment set t % 4) requests A(t+3) (slots 0, 2) and, behind its
    // sync point, X(t+8) into its own ring slot.  d = stages left after this one (a literal; kFar in the steady state): what
    // still exists to be requested, and how many of the wave's youngest operations may stay in flight at the sync point
    // (X(t+1) and A(t+1) have landed): the operations issued behind A(t+1) --
    //   stage t-2: X(t+6) [d+2 >= 8]; stage t-1: A(t+2) [d+1 >= 3], X(t+7) [d+1 >= 8]; stage t: A(t+3) [d >= 3]
    constexpr int kFar = 64;
    auto stage1 = [&](int as, int t, bool first, bool bound, bool pre, int d) __attribute__((always_inline)) {
        int nbuf = buf + 1;
        if (nbuf == 8) nbuf = 0;
        const int as3 = (as + 3) & 3;
        const bool lda = d >= 3, dmax = d >= 8;
        const int n_wait = (d + 2 >= 8 ? 2 : 0) + (d + 1 >= 3 ? 4 : 0) + (d + 1 >= 8 ? 2 : 0) + (d >= 3 ? 4 : 0);
        if (first && !I8) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) ea[rt] = ea_next[rt];
        }
        if (pre && !I8) {
            const int kb = (t >> 1) + 1;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) nsc[rt] = sc[wpiece0[rt] * kMaxKB + kb];
        }
        if (bound && !I8) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                float nm;
                split_scale(nsc[rt], ea_next[rt], nm);
                ratio[rt] = uniform_f32(mant[rt] * __builtin_amdgcn_rcpf(nm));
                mant[rt] = nm;
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
            const int tt = s2 >> 1, q = s2 & 1;
            if (tt < NTA) mma(as, s2);
            SGLK_FENCE();
            if (q == 0 && tt + 1 < NTA && !(SGLK_ABL(ABL, 32) && d >= kFar)) ld_bh(tt + 1, buf);   // the other window has been free since the previous tile's second MFMA
            if (q == 1 && bound && tt < NTA && !I8) xsv[tt] = xs_tab[((t >> 1) + 1) * kBM + tt * 32 + r32];
            // one vector-memory instruction per MFMA slot (the four waves run in step, so a burst of one wave is a burst of four
            // in front of the CU's one 64-byte-per-clock address path)
            if (s2 < 4 && lda && !(SGLK_ABL(ABL, 4) && d >= kFar)) ld_a_half(as3, s2 >> 1, t + 3, s2 & 1);
            if (bound && !I8) rescale_slot(s2);
            if (first && !I8) rescale_slot(8 + s2);
            if (s2 == 6 && d >= 1) {   // every fragment of this stage has been read
                if (SGLK_ABL(ABL, 16) && d >= kFar) __builtin_amdgcn_s_waitcnt(wc(63, 0));   // developer builds: ABL = timing ablations, wrong results
                else if (n_wait == 12) __builtin_amdgcn_s_waitcnt(wc(12, 0));        // (the builtin wants a literal)
                else if (n_wait == 10) __builtin_amdgcn_s_waitcnt(wc(10, 0));
                else if (n_wait == 8) __builtin_amdgcn_s_waitcnt(wc(8, 0));
                else if (n_wait == 4) __builtin_amdgcn_s_waitcnt(wc(4, 0));
                else __builtin_amdgcn_s_waitcnt(wc(0, 0));
                if (!(SGLK_ABL(ABL, 8) && d >= kFar)) __builtin_amdgcn_s_barrier();
                if (!(SGLK_ABL(ABL, 32) && d >= kFar)) ld_bh(0, nbuf);
            }
            if (s2 == 6 && dmax && !(SGLK_ABL(ABL, 2) && d >= kFar)) issue_x(t + 8, buf, 0);
            if (s2 == 7 && dmax && !(SGLK_ABL(ABL, 2) && d >= kFar)) issue_x(t + 8, buf, 1);
            SGLK_FENCE();
        }
        buf = nbuf;
    };
    // K block kb in phase ph (sets 2 ph, 2 ph + 1), e = K blocks behind it (a literal; kFar: at least four)
    auto kblock1 = [&](int ph, int e, int kb) __attribute__((always_inline)) {
        const int t = 2 * kb;
        if (MODE == MODE_DOWN && e == 0 && my_slot >= 0) my_tw = p.topk_weights[my_slot];   // covered by stage T-2's vmcnt(0)
        stage1(2 * ph, t, true, false, e >= 1, e >= kFar ? kFar : 2 * e + 1);
        stage1(2 * ph + 1, t + 1, false, e >= 1, false, e >= kFar ? kFar : 2 * e);
    };
    if constexpr (TERMS == 1) {
        // phases from the END of the reduction (literal set indices): the last four blocks run in phases 0 1 0 1; NMOD = kblocks % 2
        // = the phase of block 0
        ld_bh(0, 0);
        SGLK_FENCE();
        int kb = 0;
        if (NMOD == 1 && kblocks >= 5) kblock1(1, kFar, kb++);
        for (; kb + 2 <= kblocks - 4; kb += 2) {
            kblock1(0, kFar, kb);
            kblock1(1, kFar, kb + 1);
        }
        if (kblocks >= 4) kblock1(0, 3, kb++);
        if (kblocks >= 3) kblock1(1, 2, kb++);
        kblock1(0, 1, kb++);
        kblock1(1, 0, kb);
    } else {
        ld_bh(0, 0);
        ld_bl(0, 0);
        SGLK_FENCE();
        int kb = 0;
        if (NMOD == 2) kblock(2, KB_STEADY, kb++);
        if (NMOD >= 1) kblock(0, KB_STEADY, kb++);
        for (; kb + 3 <= kblocks - 2; kb += 3) {
            kblock(1, KB_STEADY, kb);
            kblock(2, KB_STEADY, kb + 1);
            kblock(0, KB_STEADY, kb + 2);
        }
        kblock(1, KB_PENULT, kb);
        kblock(2, KB_LAST, kb + 1);
    }
#undef SGLK_FENCE
    SGLK_STAMP(20);
    if (p.prio == 1) __builtin_amdgcn_s_setprio(0);
    else if (p.prio == 2) __builtin_amdgcn_s_setprio(1);
#ifdef SGLK_DEV_ABLATE
    if (p.dbg && tid == 0) {   // shader clocks / 100 MHz ticks over the main loop -> the clock the chip held
        p.dbg[32 * L + 0] = __builtin_amdgcn_s_memtime() - clk0;
        p.dbg[32 * L + 1] = p.dbg[32 * L + 20] - p.dbg[32 * L + 19];
    }
#endif
    if (MODE == MODE_DOWN && tid < kBM) tw_tab[tid] = my_tw;

    // ---- epilogue (ring dead).  32x32 accumulator: lane = token column (l & 31); register i = weight row
    //      (i & 3) + 8 * (i >> 2) + 4 * (l >> 5) of the row tile ----
    __syncthreads();
    SGLK_STAMP(25);
    int tidv = tid;
    asm volatile("" : "+v"(tidv));
    const int r32e = tidv & 31, he = (tidv >> 5) & 1;
    if constexpr (I8 && MODE == MODE_GATE_UP) {
        // int8: h = silu(gate) * up in fp32 exactly as gemm_i8_256.hip forms it ((xs * acc) * ws, separately rounded), then the
        // oracle's per-token quantisation over the WHOLE row of N columns (test_moe_int8.py:23-31,83-86): this workgroup holds
        // 128 of them, the m-tile's other n_tiles - 1 workgroups the rest.  Every workgroup adds its rows' maxima to
        // row_amax[position] (atomic max on the bit pattern of a non-negative float), then announces itself on the m-tile's
        // arrival counter and waits until all n_tiles have: the n_tiles workgroups of an m-tile are consecutive workgroups of
        // one XCD (see the kernel), dispatched in order, so whoever waits waits for workgroups that are already running or next
        // in line -- at most one m-tile per XCD is ever partly dispatched, and its waiters hold at most n_tiles - 1 slots.
        // (The wait is bounded all the same: on expiry the rows' scale becomes NaN instead of the launch hanging.)
        const float* ws_tab = sc;
        const float* xs_f = reinterpret_cast<const float*>(xs_tab);
        float v[4][16];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const float xs = xs_f[tt * 32 + r32e];
            float am = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * he;
                const float g = xs * (float)acc[0][tt][i] * ws_tab[wpiece0[0] * 16 + row];
                const float u = xs * (float)acc[1][tt][i] * ws_tab[wpiece0[1] * 16 + row];
                v[tt][i] = silu_f32(g) * u;
                am = fmaxf(am, fabsf(v[tt][i]));
            }
            am = fmaxf(am, __shfl_xor(am, 32));
            if (he == 0) amax_tab[wn * kBM + tt * 32 + r32e] = am;
        }
        __syncthreads();
        // Everything the workgroups tell each other travels in device-scope atomics (performed at the memory side, relaxed):
        // no release / acquire at agent scope, which on this chip means writing back / invalidating the XCD's whole L2 under
        // the neighbours' operand streams (measured: GEMM-1 3x slower).  The maxima are RETURNING atomics, so the counted wait
        // in front of the barrier sees them performed before thread 0 announces the workgroup.
        if (tidv < kBM && tidv < cur.rows) {
            const float am = fmaxf(fmaxf(amax_tab[tidv], amax_tab[kBM + tidv]), fmaxf(amax_tab[2 * kBM + tidv], amax_tab[3 * kBM + tidv]));
            const unsigned old = __hip_atomic_fetch_max(p.row_amax + cur.pos0 + tidv, __float_as_uint(am), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" ::"v"(old));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        SGLK_STAMP(28);
        int* flag = reinterpret_cast<int*>(smem + kRowTabOff);   // GATE_UP has no row table
        if (tidv == 0) {
            const int before = __hip_atomic_fetch_add(p.arrivals + cur.mt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int it = 0;
            if (before + 1 < p.n_tiles) {
                while (__hip_atomic_load(p.arrivals + cur.mt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.n_tiles && it < (1 << 22)) {
                    __builtin_amdgcn_s_sleep(4);
                    ++it;
                }
            }
            flag[0] = it >= (1 << 22);
        }
        __syncthreads();
        SGLK_STAMP(29);
        if (tidv < kBM) {
            unsigned u = 0u;
            if (tidv < cur.rows) u = __hip_atomic_load(p.row_amax + cur.pos0 + tidv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            amax_tab[tidv] = flag[0] ? __uint_as_float(0x7fc00000u) : __uint_as_float(u);
        }
        __syncthreads();
        SGLK_STAMP(30);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int r = tt * 32 + r32e;
            // as quant_int8_rows_f32_kernel (quant.hip): torch evaluates 127 / absmax as reciprocal(absmax) * 127
            const float am = fmaxf(amax_tab[r], p.quant_floor);
            const float inv = (1.0f / am) * 127.0f;
            if (cur.ntile == 0 && wn == 0 && he == 0 && r < cur.rows) p.out_scale_f32[cur.pos0 + r] = am / 127.0f;
            unsigned char* rowp = smem + r * 128;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                unsigned d = 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j) d |= ((unsigned)((int)rintf(v[tt][rg * 4 + j] * inv) & 0xff)) << (8 * j);
                const int pos = wn * 32 + rg * 8 + he * 4;   // natural column order (pack.hip's int8 tiles keep k in order)
                const int chunk = (pos >> 4) ^ (r & 7);
                *reinterpret_cast<unsigned*>(rowp + chunk * 16 + (pos & 15)) = d;
            }
        }
        SGLK_STAMP(26);
        __syncthreads();
        SGLK_STAMP(27);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = it * 256 + tidv;
            const int r = idx >> 3, pc = idx & 7, lc = pc ^ (r & 7);
            if (r < cur.rows) {
                const uint4 val = *reinterpret_cast<const uint4*>(smem + r * 128 + pc * 16);
                *reinterpret_cast<uint4*>((unsigned char*)p.out + (int64_t)(cur.pos0 + r) * p.out_stride + cur.ntile * 128 + lc * 16) = val;
            }
        }
    } else if constexpr (I8) {
#pragma clang fp contract(off)
        // int8 DOWN: ic2[slot] = ((xs * acc) * ws) * topk_w in bf16, the fp8 form's image and stores; PLAIN (dense
        // int8_scaled_mm, /root/reference/test_gemm_int8.py:41-47): out[row] = (xs * acc) * ws + bias, separately rounded (no
        // mul+add contraction in this block, as in gemm_i8_256.hip)
        constexpr int kRowB = 512;
        const float* ws_tab = sc;
        const float* xs_f = reinterpret_cast<const float*>(xs_tab);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int r = tt * 32 + r32e;
            unsigned char* rowp = smem + r * kRowB;
            const float tw = MODE == MODE_DOWN ? tw_tab[r] : 1.f, xs = xs_f[r];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    float o[4];
                    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (MODE == MODE_PLAIN) b4 = *reinterpret_cast<const float4*>(smem + kRowTabOff + (wn * 64 + rt * 32 + rg * 8 + he * 4) * 4);
                    const float bj[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = xs * (float)acc[rt][tt][rg * 4 + j] * ws_tab[wpiece0[rt] * 16 + rg * 8 + he * 4 + j];
                        if (MODE == MODE_DOWN) o[j] *= tw;
                        else if (p.bias) o[j] = o[j] + bj[j];   // (without a bias the oracle adds nothing: -0 stays -0)
                    }
                    uint2 val;
                    val.x = pack_bf16x2(o[0], o[1]);
                    val.y = pack_bf16x2(o[2], o[3]);
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
            const int idx = it * 256 + tidv;
            const int r = idx >> 5, pc = idx & 31, lc = pc ^ (r & 15);
            if (r < cur.rows) {
                const uint4 val = *reinterpret_cast<const uint4*>(smem + r * kRowB + pc * 16);
                *reinterpret_cast<uint4*>(outp + (int64_t)(MODE == MODE_PLAIN ? cur.pos0 + r : slot_tab[r]) * p.out_stride + cur.ntile * 256 + lc * 8) = val;
            }
        }
    } else if constexpr (MODE == MODE_GATE_UP && TERMS == 1) {
        // a8: ic1 = silu(gate) * up for this workgroup's 128 columns = ONE K block of GEMM-2, quantised like `hidden`: per-token
        // amax over the four waves, power-of-two scale, e4m3, stored in the packed-tile k order (quant_row_block128's format)
        float v[4][16];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            float am = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float g = acc[0][tt][i] * mant[0], u = acc[1][tt][i] * mant[1];
                v[tt][i] = silu_f32(g) * u;
                am = fmaxf(am, fabsf(v[tt][i]));
            }
            am = fmaxf(am, __shfl_xor(am, 32));
            if (he == 0) amax_tab[wn * kBM + tt * 32 + r32e] = am;
        }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int r = tt * 32 + r32e;
            const float am = fmaxf(fmaxf(amax_tab[r], amax_tab[kBM + r]), fmaxf(amax_tab[2 * kBM + r], amax_tab[3 * kBM + r]));
            const int sb = e8m0_for_amax(am);
            const float inv = inv_scale_of(sb);
            if (wn == 0 && he == 0 && r < cur.rows) p.out_s[(int64_t)(cur.pos0 + r) * p.out_s_stride + cur.ntile] = (uint8_t)sb;
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
            const int idx = it * 256 + tidv;
            const int r = idx >> 3, pc = idx & 7, lc = pc ^ (r & 7);
            if (r < cur.rows) {
                const uint4 val = *reinterpret_cast<const uint4*>(smem + r * 128 + pc * 16);
                *reinterpret_cast<uint4*>((unsigned char*)p.out + (int64_t)(cur.pos0 + r) * p.out_stride + cur.ntile * 128 + lc * 16) = val;
            }
        }
    } else if constexpr (MODE == MODE_GATE_UP) {
        // ic1 = bf16(silu(gate) * up) -- rounded to bf16 ONCE, as the bf16 kernel does -- for this workgroup's 128 columns = one K
        // block of GEMM-2, then split exactly like `hidden`: per-token amax over the four waves, power-of-two scale, (hi, lo),
        // stored [hi 64 | lo 64] per 64 group in the packed-tile k order
        unsigned vp[4][8];   // the tile's ic1 values as bf16 pairs (registers i, i + 1 of the accumulator)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            float am = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const float g0 = acc[0][tt][i] * mant[0], u0 = acc[1][tt][i] * mant[1];
                const float g1 = acc[0][tt][i + 1] * mant[0], u1 = acc[1][tt][i + 1] * mant[1];
                // rounding to bf16 is monotone: the largest rounded magnitude is the rounded largest magnitude -> the maximum is
                // taken over the fp32 products (one v_max3 per pair instead of two unpacks + max3) and rounded ONCE below
                const float h0 = silu_f32(g0) * u0, h1 = silu_f32(g1) * u1;
                vp[tt][i >> 1] = pack_bf16x2(h0, h1);
                am = fmaxf(am, fmaxf(fabsf(h0), fabsf(h1)));
            }
            am = __uint_as_float(pack_bf16x2(am, 0.f) << 16);
            am = fmaxf(am, __shfl_xor(am, 32));
            if (he == 0) amax_tab[wn * kBM + tt * 32 + r32e] = am;
        }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int r = tt * 32 + r32e;
            const float am = fmaxf(fmaxf(amax_tab[r], amax_tab[kBM + r]), fmaxf(amax_tab[2 * kBM + r], amax_tab[3 * kBM + r]));
            const int sb = sp_e8m0_for_amax(am);
            if (wn == 0 && he == 0 && r < cur.rows) p.out_s[(int64_t)(cur.pos0 + r) * p.out_s_stride + cur.ntile] = (uint8_t)sb;
            unsigned char* rowp = smem + r * 256;      // image row: [group 0: hi 64 | lo 64][group 1: hi 64 | lo 64]
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {           // register groups 2rp, 2rp + 1 = eight values = two dwords of hi and of lo
                float v8[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v8[2 * j] = __uint_as_float(vp[tt][rp * 4 + j] << 16);
                    v8[2 * j + 1] = __uint_as_float(vp[tt][rp * 4 + j] & 0xffff0000u);
                }
                unsigned hi[2], lo[2];
                split8(v8, sb, hi, lo);
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
            const int idx = it * 256 + tidv;
            const int r = idx >> 4, pc = idx & 15, lc = pc ^ (r & 15);
            if (r < cur.rows) {
                const uint4 val = *reinterpret_cast<const uint4*>(smem + r * 256 + pc * 16);
                *reinterpret_cast<uint4*>((unsigned char*)p.out + (int64_t)(cur.pos0 + r) * p.out_stride + cur.ntile * 256 + lc * 16) = val;
            }
        }
    } else {
        // ic2[slot] = topk_w * (acc * mant) in bf16: XOR-swizzled [token][256 columns] image (128 rows x 512 B = the whole ring),
        // whole rows out by slot
        constexpr int kRowB = 512;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int r = tt * 32 + r32e;
            unsigned char* rowp = smem + r * kRowB;
            const float tw = MODE == MODE_DOWN ? tw_tab[r] : 1.f;   // PLAIN (dense fp8_scaled_mm): out[row] = acc * scale + bias
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const float sc_w = mant[rt] * tw;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (MODE == MODE_PLAIN) b4 = *reinterpret_cast<const float4*>(smem + kRowTabOff + (wn * 64 + rt * 32 + rg * 8 + he * 4) * 4);
                    uint2 val;
                    if (MODE == MODE_PLAIN) {
                        val.x = pack_bf16x2(acc[rt][tt][rg * 4 + 0] * sc_w + b4.x, acc[rt][tt][rg * 4 + 1] * sc_w + b4.y);
                        val.y = pack_bf16x2(acc[rt][tt][rg * 4 + 2] * sc_w + b4.z, acc[rt][tt][rg * 4 + 3] * sc_w + b4.w);
                    } else {
                        val.x = pack_bf16x2(acc[rt][tt][rg * 4 + 0] * sc_w, acc[rt][tt][rg * 4 + 1] * sc_w);
                        val.y = pack_bf16x2(acc[rt][tt][rg * 4 + 2] * sc_w, acc[rt][tt][rg * 4 + 3] * sc_w);
                    }
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
            const int idx = it * 256 + tidv;
            const int r = idx >> 5, pc = idx & 31, lc = pc ^ (r & 15);
            if (r < cur.rows) {
                const uint4 val = *reinterpret_cast<const uint4*>(smem + r * kRowB + pc * 16);
                *reinterpret_cast<uint4*>(outp + (int64_t)(MODE == MODE_PLAIN ? cur.pos0 + r : slot_tab[r]) * p.out_stride + cur.ntile * 256 + lc * 8) = val;
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
#undef SGLK_STAMP
}

template <int MODE, int NMOD, int ABL, int TERMS, bool I8>
__global__ __launch_bounds__(256, 2) void moe_gemm_fp8w_s128_kernel(const A8GemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[kLds];
    // Workgroup -> tile without knowing the tile count first (the table entry and the count are fetched side by side: one memory
    // round trip less in every tile's prologue): the m-tiles are dealt to the 8 XCDs in groups of kGroup consecutive ones (about
    // one expert's worth at the headline shape), column tiles fastest inside a group, so that the column tiles of an m-tile share
    // its gathered rows and the m-tiles of an expert its weights inside one L2.  Workgroups past the last m-tile leave.
    constexpr int kGroup = kTileGroup;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    int mt;
    TileId t;
    if (MODE == MODE_PLAIN) {
        // dense rows: super-tiles of 8 m-tiles x up to 8 column tiles (one XCD's 64 resident workgroups share 8 row tiles and
        // 8 weight tiles instead of 8 and n_tiles), dealt round-robin to the XCDs, column super-tiles fastest so that an XCD's
        // next super-tile meets the same rows in its L2
        const int sn = p.n_tiles < 8 ? p.n_tiles : 8;
        const int nsn = (p.n_tiles + sn - 1) / sn;
        const int per = kGroup * sn;
        const int sl = j / per, rem = j - sl * per;
        const int st = sl * 8 + x;
        const int sm = st / nsn, sc_ = st - sm * nsn;
        const int mi = rem / sn;
        mt = sm * kGroup + mi;
        t.ntile = sc_ * sn + (rem - mi * sn);
        if (t.ntile >= p.n_tiles) return;
    } else {
        const int per = kGroup * p.n_tiles;
        const int gi = j / per, rem = j - gi * per;
        const int mi = rem / p.n_tiles;
        mt = (gi * 8 + x) * kGroup + mi;
        t.ntile = rem - mi * p.n_tiles;
    }
    t.mt = mt;
    t.L = mt * p.n_tiles + t.ntile;
    int4 ti;
    if (MODE == MODE_PLAIN) {   // dense rows: m-tile mt = rows [128 mt, 128 mt + 128) of p.dense_rows, no table
        if (mt >= p.max_mtiles) return;
        const int left = p.dense_rows - mt * kBM;
        ti = make_int4(0, mt * kBM, left < kBM ? left : kBM, 0);
    } else {
        ti = p.tile_info[mt < p.max_mtiles ? mt : 0];
        if (mt >= p.num_tiles[0]) return;
    }
    t.e = __builtin_amdgcn_readfirstlane(ti.x);
    t.pos0 = __builtin_amdgcn_readfirstlane(ti.y);
    t.rows = __builtin_amdgcn_readfirstlane(ti.z);
    const int nta = (t.rows + 31) >> 5;
    if (nta >= 4) run_tile<MODE, NMOD, ABL, 4, TERMS, I8>(p, smem, t);
    else if (nta == 3) run_tile<MODE, NMOD, ABL, 3, TERMS, I8>(p, smem, t);
    else if (nta == 2) run_tile<MODE, NMOD, ABL, 2, TERMS, I8>(p, smem, t);
    else run_tile<MODE, NMOD, ABL, 1, TERMS, I8>(p, smem, t);
}

}  // namespace gs128

bool moe_gemm_fp8w_s128_ok(int N, int K, int block_n) {
    // both reductions (K for GEMM-1, N for GEMM-2) in whole 128-wide blocks, 2 .. 32 of them; 128 ic1 columns / 256 output
    // columns per workgroup; a 32-row operand tile inside one scale block
    return K % 256 == 0 && N % 128 == 0 && K >= 256 && N >= 256 && K <= 128 * gs128::kMaxKB && N <= 128 * gs128::kMaxKB &&
           block_n % 32 == 0;
}

int launch_moe_gemm_fp8w_s128(int mode, const A8GemmParams& p_in, int max_mtiles, hipStream_t stream, int terms) {
    A8GemmParams p = p_in;
    // wave priority: GEMM-2's short main loop gains ~3 % when it outranks the co-resident workgroup's prologue / epilogue
    // (0.391 -> 0.380 ms at 16384 tokens); GEMM-1 is indifferent (profiles/r03_ab_s128_prio.txt).  SGLK_S128_PRIO=0 / 1 / 2 forces.
    p.prio = knobs().s128_prio >= 0 ? knobs().s128_prio : (mode == MODE_DOWN ? 1 : 0);
    if ((int64_t)max_mtiles * p.n_tiles == 0) return SGLK_OK;
    // groups of 8 m-tiles dealt round-robin to the 8 XCDs (see the kernel): every XCD gets the same number of workgroups
    const int64_t groups = ceil_div(max_mtiles, gs128::kTileGroup), groups_per_xcd = ceil_div(groups, 8);
    int64_t blocks = groups_per_xcd * 8 * p.n_tiles * gs128::kTileGroup;
    if (mode == MODE_PLAIN) {   // dense rows: super-tiles of 8 m-tiles x sn column tiles, dealt to the XCDs (see the kernel)
        const int64_t sn = p.n_tiles < 8 ? p.n_tiles : 8, supers = groups * ceil_div(p.n_tiles, sn);
        blocks = ceil_div(supers, 8) * 8 * (gs128::kTileGroup * sn);
    }
    if (p.max_mtiles != max_mtiles) SGLK_FAIL(SGLK_ERR_INVALID, "moe_gemm_fp8w_s128: max_mtiles not set in the parameter block");
    const int kblocks = p.C >> 7;
    if (p.C % 128 != 0 || kblocks < 2 || kblocks > gs128::kMaxKB)
        SGLK_FAIL(SGLK_ERR_SHAPE, "moe_gemm_fp8w_s128: reduction length %d (needs 2..%d whole 128-wide K blocks)", p.C, gs128::kMaxKB);
    if (terms != 0 && p.block_n % 32 != 0) SGLK_FAIL(SGLK_ERR_SHAPE, "moe_gemm_fp8w_s128: block_n %d is not a multiple of 32", p.block_n);
    if (terms != 0 && (p.xs_stride % 4 != 0 || ((uintptr_t)p.xs % 4) != 0)) SGLK_FAIL(SGLK_ERR_INVALID, "moe_gemm_fp8w_s128: scale rows must be 4-byte aligned");
    if (terms == 0 && (!p.x_scale_f32 || (mode == MODE_GATE_UP && (!p.out_scale_f32 || !p.row_amax || !p.arrivals))))
        SGLK_FAIL(SGLK_ERR_INVALID, "moe_gemm_fp8w_s128: int8 needs the per-row scale tables and the row-maximum exchange buffers");
    if (mode != MODE_GATE_UP && mode != MODE_DOWN && !(mode == MODE_PLAIN && terms != 1)) SGLK_FAIL(SGLK_ERR_INVALID, "moe_gemm_fp8w_s128: mode %d", mode);
    const int nmod = terms == 2 ? (kblocks - 2) % 3 : kblocks % 2;   // see run_tile: the rotation of the fragment sets
    if (terms < 0 || terms > 2) SGLK_FAIL(SGLK_ERR_INVALID, "moe_gemm_fp8w_s128: terms %d", terms);
#define SGLK_LAUNCH_S128(M_, N_, A_)                                                                                      \
    do {                                                                                                                  \
        if (terms == 2)                                                                                                   \
            hipLaunchKernelGGL((gs128::moe_gemm_fp8w_s128_kernel<M_, N_, A_, 2, false>), dim3((unsigned)blocks), dim3(256), 0, stream, p); \
        else if (terms == 1)                                                                                              \
            hipLaunchKernelGGL((gs128::moe_gemm_fp8w_s128_kernel<M_, (N_) & 1, 0, 1, false>), dim3((unsigned)blocks), dim3(256), 0, stream, p);  \
        else                                                                                                              \
            hipLaunchKernelGGL((gs128::moe_gemm_fp8w_s128_kernel<M_, (N_) & 1, 0, 1, true>), dim3((unsigned)blocks), dim3(256), 0, stream, p);   \
    } while (0)
#ifdef SGLK_DEV_ABLATE   // developer-only timing ablations (wrong results by design)
    const int abl = knobs().rescale_ablate;
#define SGLK_LAUNCH_S128_B(M_, N_)                                                                 \
    do {                                                                                           \
        switch (abl) {                                                                             \
            case 1: SGLK_LAUNCH_S128(M_, N_, 1); break;                                            \
            case 2: SGLK_LAUNCH_S128(M_, N_, 2); break;                                            \
            case 4: SGLK_LAUNCH_S128(M_, N_, 4); break;                                            \
            case 7: SGLK_LAUNCH_S128(M_, N_, 7); break;                                            \
            default: SGLK_LAUNCH_S128(M_, N_, 0); break;                                           \
        }                                                                                          \
    } while (0)
    // one-term int8 GEMM-1: 2 no X DMA, 4 no weight loads, 8 no barrier, 16 no counted wait, 32 no fragment reads (steady state)
#define SGLK_LAUNCH_S128_I8ABL(N_, A_) \
    hipLaunchKernelGGL((gs128::moe_gemm_fp8w_s128_kernel<MODE_GATE_UP, N_, A_, 1, true>), dim3((unsigned)blocks), dim3(256), 0, stream, p)
    if (terms == 0 && mode == MODE_GATE_UP && abl >= 100) {
        const int a2 = abl - 100;
        if (nmod == 0) {
            switch (a2) {
                case 2: SGLK_LAUNCH_S128_I8ABL(0, 2); break;
                case 4: SGLK_LAUNCH_S128_I8ABL(0, 4); break;
                case 8: SGLK_LAUNCH_S128_I8ABL(0, 8); break;
                case 24: SGLK_LAUNCH_S128_I8ABL(0, 24); break;
                case 32: SGLK_LAUNCH_S128_I8ABL(0, 32); break;
                case 62: SGLK_LAUNCH_S128_I8ABL(0, 62); break;
                default: SGLK_LAUNCH_S128_I8ABL(0, 0); break;
            }
            SGLK_CHECK_LAUNCH("moe_gemm_fp8w_s128");
            return SGLK_OK;
        }
    }
#undef SGLK_LAUNCH_S128_I8ABL
#else
#define SGLK_LAUNCH_S128_B(M_, N_) SGLK_LAUNCH_S128(M_, N_, 0)
#endif
    if (mode == MODE_GATE_UP) {
        if (nmod == 0) SGLK_LAUNCH_S128_B(MODE_GATE_UP, 0);
        else if (nmod == 1) SGLK_LAUNCH_S128_B(MODE_GATE_UP, 1);
        else SGLK_LAUNCH_S128_B(MODE_GATE_UP, 2);
    } else if (mode == MODE_DOWN) {
        if (nmod == 0) SGLK_LAUNCH_S128_B(MODE_DOWN, 0);
        else if (nmod == 1) SGLK_LAUNCH_S128_B(MODE_DOWN, 1);
        else SGLK_LAUNCH_S128_B(MODE_DOWN, 2);
    } else {   // dense rows (one "expert"): two-term fp8 or int8
#define SGLK_LAUNCH_S128_P(N_)                                                                                                         \
    do {                                                                                                                                \
        if (terms == 2)                                                                                                                 \
            hipLaunchKernelGGL((gs128::moe_gemm_fp8w_s128_kernel<MODE_PLAIN, N_, 0, 2, false>), dim3((unsigned)blocks), dim3(256), 0, stream, p); \
        else                                                                                                                            \
            hipLaunchKernelGGL((gs128::moe_gemm_fp8w_s128_kernel<MODE_PLAIN, (N_) & 1, 0, 1, true>), dim3((unsigned)blocks), dim3(256), 0, stream, p); \
    } while (0)
        if (nmod == 0) SGLK_LAUNCH_S128_P(0);
        else if (nmod == 1) SGLK_LAUNCH_S128_P(1);
        else SGLK_LAUNCH_S128_P(2);
#undef SGLK_LAUNCH_S128_P
    }
#undef SGLK_LAUNCH_S128_B
#undef SGLK_LAUNCH_S128
    SGLK_CHECK_LAUNCH("moe_gemm_fp8w_s128");
    return SGLK_OK;
}

}  // namespace sglk
