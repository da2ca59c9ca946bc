// Large-M grouped W8A16 GEMM of fused_experts: 256 tokens x 256 weight rows per workgroup, 8 waves, 3-deep LDS-DMA ring,
// mfma_f32_32x32x16_bf16 variant (half the MFMA issue slots of the 16x16x32 kernel for the same math; the packed
// 16-row weight tiles are re-mapped to the 32-row operand when they are read from LDS, the pack order is unchanged).
//
// Same math contract as moe_gemm_fp8w.hip (oracle: /root/reference/test_moe_fp8_ext.py:22-25,70-91); this is the
// kernel fused_experts picks when experts receive >= ~192 rows (prefill / the M=16384 headline).
//
// Structure (MI355X: 1 workgroup of 512 threads per CU = 2 waves per SIMD, 144 KiB of the 160 KiB LDS):
//   * waves 4(n) x 2(m): each wave owns 64 weight rows x 128 tokens = 4 x 8 tiles of mfma_f32_16x16x32_bf16
//     (weights = A operand, tokens = B operand; 128 accumulator registers per lane);
//   * K is walked in 64-deep stages through a ring of THREE LDS buffers (X 32 KiB bf16 + W 16 KiB fp8 each).
//     Loads run two stages ahead: per stage ONE counted `s_waitcnt vmcnt(6)` (the 6 = this wave's LDS-DMA
//     instructions per stage, i.e. the next stage stays in flight), ONE raw s_barrier, then the DMA for stage t+2 is
//     issued into the buffer everybody finished reading in stage t-1, then 64 MFMAs;
//   * block scales without a second accumulator: scale s = m * 2^e.  2^e is applied for free (and exactly) by
//     v_cvt_scalef32_pk_bf16_fp8, the mantissa m in +-[1,2) is carried by keeping the accumulator in units of the
//     current block's m:  T <- T * (m_prev / m_cur) at every 128-wide K block (one v_mul per accumulator register,
//     ratio always finite), C = m_last * T.  Products stay exact; the extra error is one fp32 rounding per K block;
//   * epilogue through LDS (ring is dead by then): XOR-swizzled [token][column] image, read back as whole rows and
//     stored with 16-byte accesses (ic1 rows are contiguous positions, ic2 rows are scattered by slot).
#include <stdlib.h>

#include <type_traits>

#include "knobs.h"
#include "moe_internal.h"

namespace sglk {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

namespace g256i {

#ifdef SGLK_NO_TAIL_SKIP   // A/B build: tail tiles run every MFMA like full tiles
constexpr bool kTailSkip = false;
#else
constexpr bool kTailSkip = true;
#endif
constexpr int kBM = 256;
constexpr int kStageX = kBM * 128;        // 32 KiB
constexpr int kStageW = 16 * 1024;        // 16 KiB: 16 packed 16x64 fp8 tiles
constexpr int kStage = kStageX + kStageW; // 48 KiB
constexpr int kRing = 3;
constexpr int kScaleOff = kRing * kStage; // 144 KiB, then the scale table
constexpr int kMaxKBlocks = 64;           // reduction length <= 8192
constexpr int kRowTabOff = kScaleOff + 16 * kMaxKBlocks * 4;   // + 4 KiB, then the per-row tables of the DOWN epilogue
constexpr int kTicketOff = kRowTabOff + 2 * kBM * 4;    // + 2 KiB: output slot and routing weight per tile row
constexpr int kLds = kTicketOff + 16;                   // + the next dynamic tile ticket of the workgroup

SGLK_DEV void glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0); }

SGLK_DEV bf16x8 cvt8(unsigned lo, unsigned hi, float pow2) {
    const bf16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, pow2, false);
    const bf16x2 b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, pow2, true);
    const bf16x2 c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, pow2, false);
    const bf16x2 d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, pow2, true);
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1];
    r[4] = c[0]; r[5] = c[1]; r[6] = d[0]; r[7] = d[1];
    return r;
}

// s = mant * pow2 with pow2 = 2^floor(log2|s|) (0 for zero/denormal s), mant in +-[1,2) (1 when pow2 == 0;
// inf/nan pass through in mant so they still poison the result)
SGLK_DEV float uniform_f32(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// scales are wave-uniform: keep everything derived from them in SGPRs (the accumulators own the VGPR budget)
SGLK_DEV void split_scale(float s_in, float& pow2, float& mant) {
    const float s = uniform_f32(s_in);
    const unsigned u = __float_as_uint(s);
    const unsigned ex = u & 0x7f800000u;
    const bool tiny = ex == 0u, special = ex == 0x7f800000u;
    pow2 = tiny ? 0.f : (special ? 1.f : __uint_as_float(ex));
    mant = tiny ? 1.f : (special ? s : __uint_as_float((u & 0x807fffffu) | 0x3f800000u));
}

// NRT x NTT = the wave's tile in 32-row weight tiles x 32-token tiles (8 accumulator tiles either way):
//   NRT = 2: waves 4(n) x 2(m), 64 weight rows x 128 tokens  (LDS reads per stage: X 128 KiB + W 32 KiB)
//   NRT = 4: waves 2(n) x 4(m), 128 weight rows x 64 tokens  (X 64 KiB + W 64 KiB: 20 % fewer LDS bytes -- the fp8
//            weight fragments are half the size of the bf16 token fragments -- for twice the conversions)
template <int MODE, int RESCALE, int NRT>
__global__ __launch_bounds__(512, 2) void moe_gemm_fp8w_256i_kernel(const MoeGemmParams p) {
    constexpr int NTT = 8 / NRT;
    // MFMA slot of a k-step that carries the first of its two LDS-DMA pieces (the second rides in slot 7): 3 = beside the
    // last operand read, 5 = a conversion-only gap
#ifdef SGLK_DMA_SLOT_A
    constexpr int kDmaSlotA = SGLK_DMA_SLOT_A;
#else
    constexpr int kDmaSlotA = 3;
#endif
    __shared__ __attribute__((aligned(16))) unsigned char smem[kLds];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = NRT == 2 ? (wave & 3) : (wave & 1), wm = NRT == 2 ? (wave >> 2) : (wave >> 1);

    // ---- persistent workgroups: the launch has at most one workgroup per CU and each walks a strided list of tiles.
    // Workgroups are dealt round-robin over the 8 XCDs; XCD x owns the contiguous tile range [xs, xs + xl) (equal
    // shares, bijective for any count) so that one expert's weights and one m-tile's activations stay in one L2, and
    // the workgroups of an XCD take consecutive tiles of that range in every round.
    const int nmt = p.num_tiles[0];
    const int live = nmt * p.n_tiles;
    // linear tile id -> (m-tile, column tile).  Default: column tiles fastest (the six workgroups of an m-tile are
    // neighbours and share its gathered token rows).  SGLK_TILE_ORDER_N_MAJOR (A/B build): m-tiles fastest, so the m-tiles of
    // one expert that stream the same weight slab are neighbours instead.
#ifdef SGLK_TILE_ORDER_N_MAJOR
#define SGLK_SPLIT_L(Lx, mt_, nt_) const int nt_ = (Lx) / nmt, mt_ = (Lx) - nt_ * nmt
#define SGLK_MT_OF(Lx) ((Lx) % nmt)
#else
#define SGLK_SPLIT_L(Lx, mt_, nt_) const int mt_ = (Lx) / p.n_tiles, nt_ = (Lx) - mt_ * p.n_tiles
#define SGLK_MT_OF(Lx) ((Lx) / p.n_tiles)
#endif
    int xs, xl, nbx;
    {
        // a launch of fewer than 8 workgroups (grid-cap knob of the tests) cuts the tiles into that many ranges instead
        const int np = (int)gridDim.x < 8 ? (int)gridDim.x : 8;
        const int x = np == 8 ? (int)(blockIdx.x & 7) : (int)blockIdx.x, q = live / np, r = live - q * np;
        xs = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        xl = q + (x < r ? 1 : 0);
        nbx = ((int)gridDim.x - x + np - 1) / np;   // workgroups of this launch on XCD x
    }
    // Tile sequence of the workgroup inside its XCD's range: four static rounds (j0 + k * nbx), then tickets from a
    // per-XCD counter (p.tickets, zeroed by the caller) -- a workgroup that drew cheap tail tiles simply draws again, like
    // hardware dispatch would, while the table prefetch below keeps its one-tile lookahead.  A ticket is requested
    // after one main loop and published (LDS) after the next, so its latency never shows.
    int jt = blockIdx.x >> 3;
    if (jt >= xl) return;
    int jt_n = jt + nbx, jt_nn = jt + 2 * nbx;
    int* ticket_lds = reinterpret_cast<int*>(smem + kTicketOff);
    int my_ticket = -1;

    // What a tile needs before its first LDS-DMA can go out sits behind two dependent round trips (tile table ->
    // row table / scales).  For every tile but the first they are taken during the PREVIOUS tile: the table entry is
    // loaded at its start, the dependent rows and scales just before its epilogue, so the next tile opens with
    // nothing but the operand DMA to wait for.
    struct Meta {
        int4 ti;              // {expert, first position, rows, -}
        int slots[4];         // GATE_UP: sorted_slot of the wave's four DMA row groups
        int my_slot;          // DOWN: sorted_slot of tile row tid (tid < 256) ...
        float tw;             // ... and its routing weight (third round trip: fetched after the epilogue's stores)
        float sc_reg[2];      // scale-table entries sc[tid], sc[tid + 512]  (sc[piece * 64 + kb])
    };
    const int kblocks_ = p.C >> 7;
    auto fetch_meta = [&](int Lq, Meta& m) __attribute__((always_inline)) {
        SGLK_SPLIT_L(Lq, mt, nt);
        (void)mt;
        const int e_ = __builtin_amdgcn_readfirstlane(m.ti.x), pos0_ = __builtin_amdgcn_readfirstlane(m.ti.y);
        const int rows_ = __builtin_amdgcn_readfirstlane(m.ti.z);
        const bool ksp = MODE == MODE_PLAIN && p.ksplit > 1;      // dense split-K: the "expert" is the K range
        const float* scale_e = ksp ? p.w_scale + (int64_t)e_ * p.split_kblocks : p.w_scale + (int64_t)e_ * p.scale_rows * p.scale_cols;
        const float inv_bn = 1.0f / (float)p.block_n;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * 512;                 // table entry sc[piece * 64 + kb]: no division by kblocks
            const int piece = i >> 6, kb = i & (kMaxKBlocks - 1);
            m.sc_reg[j] = 0.f;
            if (kb < kblocks_) {
                int row16;
                if (MODE == MODE_GATE_UP) row16 = (piece < 8) ? nt * 8 + piece : (p.n_half >> 4) + nt * 8 + (piece - 8);
                else row16 = nt * 16 + piece;
                // floor(row / block_n) through one float multiply: exact for rows < 2^20 (the +0.5 keeps the product
                // at least 0.5 / block_n away from an integer, far more than the rounding error)
                const int srow = (int)(((float)(row16 * 16) + 0.5f) * inv_bn);
                m.sc_reg[j] = scale_e[srow * p.scale_cols + kb];
            }
        }
        m.my_slot = -1;
        m.tw = 0.f;
        if (MODE == MODE_DOWN && tid < rows_) m.my_slot = p.sorted_slot[pos0_ + tid];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            m.slots[i] = 0;
            if (MODE == MODE_GATE_UP) {
                const int r = (wave * 4 + i) * 8 + (lane >> 3);
                m.slots[i] = p.sorted_slot[pos0_ + (r < rows_ ? r : 0)];
            }
        }
    };
    Meta cur, nxt;
    cur.ti = p.tile_info[SGLK_MT_OF(xs + jt)];
    fetch_meta(xs + jt, cur);
    if (MODE == MODE_DOWN && cur.my_slot >= 0) cur.tw = p.topk_weights[cur.my_slot];
    nxt = cur;

    for (;;) {   // ---- one tile per iteration ---------------------------------------------------------------------
#ifdef SGLK_DEV_ABLATE
    const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int L = xs + jt;
    const bool has_next_tile = jt_n < xl;
    // next tile's table entry by SCALAR load (uniform address, constant address space): lands in SGPRs, costs no
    // VGPR across the main loop and is consumed after it
    if (has_next_tile) {
        const __attribute__((address_space(4))) int* tp = reinterpret_cast<const __attribute__((address_space(4))) int*>(
            reinterpret_cast<uintptr_t>(p.tile_info + SGLK_MT_OF(xs + jt_n)));
        nxt.ti.x = tp[0];
        nxt.ti.y = tp[1];
        nxt.ti.z = tp[2];
    }
    SGLK_SPLIT_L(L, mtile, ntile);
    (void)mtile;
    const int e = __builtin_amdgcn_readfirstlane(cur.ti.x);
    const int pos0 = __builtin_amdgcn_readfirstlane(cur.ti.y);
    const int rows = __builtin_amdgcn_readfirstlane(cur.ti.z);

    const int ctiles = p.C >> 6;
    const int kblocks = p.C >> 7;
    const int T = ctiles;

    // workgroup's 16 weight row-tiles: GATE_UP = 8 gate + 8 up, DOWN = 16 consecutive
    auto piece_row16 = [&](int piece) __attribute__((always_inline)) {
        if (MODE == MODE_GATE_UP) return (piece < 8) ? ntile * 8 + piece : (p.n_half >> 4) + ntile * 8 + (piece - 8);
        return ntile * 16 + piece;
    };

    // ---- prologue loads.  Everything the tile needs besides the operands is FETCHED here but parked in registers;
    //      the LDS copies are written after the first LDS-DMA stages have been issued, so no operand load waits
    //      behind these round trips ---------------------------------------------------------------------------------
    float* sc = reinterpret_cast<float*>(smem + kScaleOff);            // sc[piece][kb]
    int* slot_tab = reinterpret_cast<int*>(smem + kRowTabOff);         // DOWN epilogue: output row (slot) ...
    float* tw_tab = reinterpret_cast<float*>(smem + kRowTabOff + kBM * 4);   // ... and routing weight per tile row
    const float sc_reg[2] = {cur.sc_reg[0], cur.sc_reg[1]};           // 16 * kblocks <= 1024 table entries
    const int my_slot = cur.my_slot;

    // ---- LDS-DMA sources: buffer descriptors (SGPRs) + ONE 32-bit per-lane offset per piece, fixed for the whole
    //      tile; the stage offset goes in the scalar soffset, so a stage costs 6 buffer_load...lds and no VALU ----
    const unsigned xbytes = (unsigned)__builtin_amdgcn_readfirstlane((int)p.x_bytes);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, xbytes, 0x00020000);
    const bool ksp = MODE == MODE_PLAIN && p.ksplit > 1;          // dense split-K: tile-table "expert" e = K range
    const unsigned char* wexp = p.w + (int64_t)e * p.w_expert_stride;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)wexp, 0, ksp ? (unsigned)(p.w_bytes_total - (int64_t)e * p.w_expert_stride) : (unsigned)p.w_expert_stride, 0x00020000);
    const unsigned x_koff = ksp ? (unsigned)e * (unsigned)p.C * 2u : 0u;   // bytes into every row of x
    unsigned xsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int rr = r < rows ? r : 0;
        int64_t xrow;
        if (MODE == MODE_GATE_UP) {
            xrow = (int64_t)(cur.slots[i] / p.topk) * p.x_stride;
        } else {
            xrow = (int64_t)(pos0 + rr) * p.x_stride;
        }
        xsrc[i] = (unsigned)(xrow * 2) + x_koff + (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) << 4);   // image swizzle: chunk ^ ((row>>1)&7)
        // rows past the tile's last one: an offset outside the descriptor's range.  The load still counts in vmcnt
        // (the stage waits are literals) but fetches nothing, so a tail tile with 40 rows does not pull 256 rows
        // through L2; whatever the LDS rows then hold only reaches accumulator columns that are never stored.
        if (r >= rows) xsrc[i] = xbytes;
    }
    unsigned wsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) wsrc[i] = (unsigned)(piece_row16(wave * 2 + i) * (ksp ? p.c_full >> 6 : ctiles)) * 1024u + lane * 16;

    // one of the wave's six 1-KiB LDS-DMA pieces of stage kt (0..3: X rows, 4..5: packed W tiles).  The pieces of a
    // stage are issued ONE PER MFMA GROUP, never as a burst: a buffer_load...lds costs the issuing wave 60-180 cycles,
    // which hides behind the MFMAs already queued on the matrix pipe but stalls the wave when six come back to back.
    auto issue_piece = [&](int kt, int buf, int i) __attribute__((always_inline)) {
        if (SGLK_ABL(RESCALE, 64) && kt > 2) return;   // bit 6: timing ablation, no LDS-DMA in the steady state
        unsigned char* sx = smem + buf * kStage;
        if (i < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lptr_t)(sx + (wave * 4 + i) * 1024), 16, xsrc[i], kt * 128, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lptr_t)(sx + kStageX + (wave * 2 + i - 4) * 1024), 16,
                                                     wsrc[i - 4], kt * 1024, 0, 0);
    };
    auto issue_stage = [&](int kt, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 6; ++i) issue_piece(kt, buf, i);
    };

    // ---- operand addressing for mfma_f32_32x32x16_bf16 (A = weights, B = tokens) ---------------------------------------
    // lane l: h = l>>5 (k octet inside the 16-wide k-step), r = l&31 (operand row / column).
    // Weight row tile rt (32 rows) = two packed 16-row pieces; octet o = 2*ks + h of row r sits in piece (r>>4) at lane
    // slot ((o&3)*16 + (r&15)), byte (o>>2)*8 (pack.hip order) -> one ds_read_b64 per (rt, ks).
    const int h = lane >> 5, r32 = lane & 31;
    int wbase[NRT];      // byte offset of (rt, lane) inside a W stage, without the k-step term
    int wpiece0[NRT];    // first piece of the row tile (for the scale table)
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
        // GATE_UP: the first NRT/2 row tiles are gate rows, the rest the matching up rows (same output columns)
        if (MODE == MODE_GATE_UP) wpiece0[rt] = (rt < NRT / 2) ? wn * NRT + rt * 2 : 8 + wn * NRT + (rt - NRT / 2) * 2;
        else wpiece0[rt] = wn * NRT * 2 + rt * 2;
        wbase[rt] = (wpiece0[rt] + (r32 >> 4)) * 1024 + (r32 & 15) * 16;
    }
    // k-step ks: octet o = 2ks + h -> slot group (o&3), half (o>>2)
    auto woff = [&](int rt, int ks) __attribute__((always_inline)) { return wbase[rt] + (((2 * ks + h) & 3) * 16) * 16 + ((2 * ks + h) >> 2) * 8; };
    // token tile tt (32 tokens): row = wm*NTT*32 + tt*32 + r32, chunk 2ks + h, swizzled by (row>>1)&7
    const int xrow0 = wm * (NTT * 32) + r32;
    auto xoff = [&](int tt, int ks) __attribute__((always_inline)) {
        const int row = xrow0 + tt * 32;
        return row * 128 + (((2 * ks + h) ^ ((row >> 1) & 7)) << 4);
    };

    f32x16 acc[NRT][NTT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][tt][i] = 0.f;

    float pow2[NRT], pow2_next[NRT], mant[NRT], ratio[NRT];

    // prologue: stages 0 and 1 complete and the first two pieces of stage 2 in flight (the rest of a stage's pieces are
    // issued from inside the main loop, two stages ahead of their use)
    issue_stage(0, 0);
    issue_stage(1, 1);
    if (T > 2) { issue_piece(2, 2, 0); issue_piece(2, 2, 1); }
    {
        sc[tid] = sc_reg[0];
        sc[tid + 512] = sc_reg[1];
        if (MODE == MODE_DOWN && tid < kBM) {
            slot_tab[tid] = my_slot;
            tw_tab[tid] = cur.tw;
        }
    }
    // stage 0 has landed (its six pieces are the oldest; stage 1 and the two pieces of stage 2 stay in flight), tables
    // visible
    asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
        split_scale(sc[wpiece0[rt] * kMaxKBlocks], pow2[rt], mant[rt]);
        pow2_next[rt] = pow2[rt];
        ratio[rt] = 1.f;
    }

    // ---- main loop ----------------------------------------------------------------------------------------------------
    // A stage = 4 k-steps of 8 MFMAs (2 weight row tiles x 4 token tiles).  One wave alone can keep the matrix pipe
    // busy (tools/probe/mfma_peak.hip: 32.4 cycles per back-to-back mfma_32x32x16 = the pipe rate) but only if nothing
    // else holds up its issue: every MFMA leaves a ~32-cycle shadow in which the same wave may issue other work for
    // free, and whatever does not fit in the shadows is pipe idle time that the SIMD's second wave only partly fills
    // (oldest-first arbitration).  So the feed of k-step c+1 -- 2 weight reads, 4 token reads, 8 conversions, the
    // LDS-DMA pieces and, at K-block boundaries, the accumulator rescale -- is cut into eight slices and slice s is
    // issued right behind MFMA s of k-step c.  The scheduling fences pin that order (left alone the scheduler
    // clusters the MFMAs and sinks every LDS read to just in front of its use).
    //   slot:      0          1          2      3          4      5      6      7
    //   feed:    W rt0+X tt0  W rt1+X tt1  X tt2  X tt3+DMA  2 cvt  2 cvt  2 cvt  2 cvt + DMA
    // Registers: converted weights wf[parity][rt], token fragments xf[parity][tt] (parity = k-step & 1), raw octets
    // wraw[rt] of the k-step being converted.
#define SGLK_FENCE() __builtin_amdgcn_sched_barrier(0)
    // token tiles (32 tokens) of this wave that hold at least one row: NTT except in an expert's last (tail) tile
    int nta = (rows - wm * (NTT * 32) + 31) >> 5;
    nta = __builtin_amdgcn_readfirstlane(nta < 0 ? 0 : (nta > NTT ? NTT : nta));
    u32x4 wfw[2][NRT], xf[2][NTT];
    u32x2 wraw[NRT];
    // MFMA slot s of a k-step -> (row tile, token tile); consecutive slots alternate accumulators
    auto slot_rt = [&](int s2) __attribute__((always_inline)) { return NRT == 2 ? (s2 >> 1) & 1 : s2 >> 1; };
    auto slot_tt = [&](int s2) __attribute__((always_inline)) { return NRT == 2 ? (s2 & 1) + 2 * (s2 >> 2) : s2 & 1; };
    auto ld_w = [&](int rt, int fbuf, int ks) __attribute__((always_inline)) {
        wraw[rt] = *reinterpret_cast<const u32x2*>(smem + fbuf * kStage + kStageX + woff(rt, ks));
    };
    // chk (a literal at every call site): the tile is a TAIL tile whose last rows end inside this wave's token range -- token
    // tiles (32 tokens) that hold no row at all are skipped: no fragment read, no MFMA, no rescale (nta = tiles with rows)
    auto ld_x = [&](int par, int tt, int fbuf, int ks, bool chk) __attribute__((always_inline)) {
        if (chk && tt >= nta) return;
        if (SGLK_ABL(RESCALE, 16) && (ks | tt)) return;   // timing ablation: one X read per stage
        xf[par][tt] = *reinterpret_cast<const u32x4*>(smem + fbuf * kStage + xoff(tt, ks));
    };
    // words 2*half, 2*half+1 of the converted row tile rt (octet low / high dword of the raw pair)
    auto cvt2 = [&](int par, int rt, int half, float sc2) __attribute__((always_inline)) {
        const unsigned src = half ? wraw[rt][1] : wraw[rt][0];
        if (SGLK_ABL(RESCALE, 8)) { wfw[par][rt][2 * half] = src; wfw[par][rt][2 * half + 1] = src; return; }   // timing ablation: no conversion
        wfw[par][rt][2 * half] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(src, sc2, false));
        wfw[par][rt][2 * half + 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(src, sc2, true));
    };
    auto mma = [&](int par, int s2, bool chk) __attribute__((always_inline)) {
        const int rt = slot_rt(s2), tt = slot_tt(s2);
        if (chk && tt >= nta) return;
        acc[rt][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfw[par][rt]),
                                                              __builtin_bit_cast(bf16x8, xf[par][tt]), acc[rt][tt], 0, 0, 0);
    };
    auto rescale = [&](int a, bool chk) __attribute__((always_inline)) {   // accumulator of MFMA slot a into units of the next K block's mantissa
        if (SGLK_ABL(RESCALE, 4)) return;   // timing ablation: no rescale
        const int rt = slot_rt(a), tt = slot_tt(a);
        if (chk && tt >= nta) return;
        // one plain v_mul_f32 per register: beside MFMAs a packed v_pk_mul_f32 costs the wave ~3x the issue time of the
        // two scalar multiplies it replaces (MI355X_MICROARCH.md, cycle constants), and the vector form of this
        // statement is always lowered to the packed instruction
#pragma unroll
        for (int i = 0; i < 16; ++i) asm("v_mul_f32 %0, %1, %0" : "+v"(acc[rt][tt][i]) : "s"(ratio[rt]));
    };
    auto sync_point = [&](bool wait6) __attribute__((always_inline)) {
        if (wait6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (SGLK_ABL(RESCALE, 1024)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // timing ablation: NO barrier (racy)
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    int buf = 0;
    // k-step ks of the stage in ring slot `buf`; every flag is a literal at the call site.
    //   fetch : read + convert the operands of the next k-step (same stage, or k-step 0 of the next stage when ks == 3)
    //   sync  : ks == 3 only -- the next stage has landed and everybody is done with this stage's buffer (S_t)
    //   dma_a / dma_b : LDS-DMA piece numbers issued in slots 3 / 7 (-1: none) for stage `dma_stage` into `dma_buf`
    //   resc_lo / resc_hi : K-block boundary -- rescale accumulators 0..4 in slots 3..7 / accumulators 5..7 in slots 0..2
    //   next_pow2 : the conversions of this k-step belong to the next K block
    auto kstep = [&](int ks, bool fetch, bool sync, bool wait6, int dma_a, int dma_b, int dma_stage, int dma_buf,
                     bool resc_lo, bool resc_hi, bool next_pow2, bool chk) __attribute__((always_inline)) {
        const int par = ks & 1, npar = par ^ 1;
        int nbuf = buf + 1;
        if (nbuf == kRing) nbuf = 0;
        const int fbuf = (ks == 3) ? nbuf : buf;
        const int fks = (ks + 1) & 3;
        // feed slices of the NEXT k-step (fks of fbuf), by slot:
        //   NRT = 2:  0: W0+X0   1: W1+X1   2: X2   3: X3        4..7: one half row tile converted per slot
        //   NRT = 4:  0: W0+X0   1: W1+X1   2: W2   3: W3        4..7: one whole row tile converted per slot
        auto feed_reads = [&](int s2) __attribute__((always_inline)) {
            if (!fetch) return;
            if (s2 < NRT) ld_w(s2, fbuf, fks);
            if (s2 < NTT) ld_x(npar, s2, fbuf, fks, chk);
        };
        auto feed_cvt = [&](int s2) __attribute__((always_inline)) {   // s2 in 4..7
            if (!fetch) return;
            if (NRT == 2) {
                const int rt = (s2 - 4) >> 1, half = (s2 - 4) & 1;
                cvt2(npar, rt, half, next_pow2 ? pow2_next[rt] : pow2[rt]);
            } else {
                const int rt = s2 - 4;
                const float sc2 = next_pow2 ? pow2_next[rt] : pow2[rt];
                cvt2(npar, rt, 0, sc2);
                cvt2(npar, rt, 1, sc2);
            }
        };
        // slot 0
        mma(par, 0, chk);
        SGLK_FENCE();
        if (sync) {
            mma(par, 1, chk);
            SGLK_FENCE();
            sync_point(wait6);
            feed_reads(0);
            feed_reads(1);
            if (resc_hi) { rescale(5, chk); }
            SGLK_FENCE();
        } else {
            feed_reads(0);
            if (resc_hi) rescale(5, chk);
            SGLK_FENCE();
            // slot 1
            mma(par, 1, chk);
            SGLK_FENCE();
            feed_reads(1);
            if (resc_hi) rescale(6, chk);
            SGLK_FENCE();
        }
        // slot 2
        mma(par, 2, chk);
        SGLK_FENCE();
        feed_reads(2);
        if (resc_hi) rescale(7, chk);
        SGLK_FENCE();
        // slot 3
        mma(par, 3, chk);
        SGLK_FENCE();
        feed_reads(3);
        if (kDmaSlotA == 3 && dma_a >= 0) issue_piece(dma_stage, dma_buf, dma_a);
        if (resc_lo) rescale(0, chk);
        SGLK_FENCE();
        // slots 4..7
#pragma unroll
        for (int s2 = 4; s2 < 8; ++s2) {
            mma(par, s2, chk);
            SGLK_FENCE();
            feed_cvt(s2);
            if (s2 == kDmaSlotA && dma_a >= 0) issue_piece(dma_stage, dma_buf, dma_a);
            if (s2 == 7 && dma_b >= 0) issue_piece(dma_stage, dma_buf, dma_b);
            if (resc_lo) rescale(s2 - 3, chk);
            SGLK_FENCE();
        }
    };
    // stage t.  first / closing: first / second stage of a 128-wide K block.  dma_mid: pieces 2..5 of stage t+2 go out in
    // k-steps 0,1; dma_tail: pieces 0,1 of stage t+3 in k-step 3 (into this stage's own buffer, free after S_t).
    auto stage = [&](int t, bool first, bool closing, bool has_next, bool wait6, bool dma_mid, bool dma_tail, bool boundary, bool chk) __attribute__((always_inline)) {
        int pbuf = buf - 1;
        if (pbuf < 0) pbuf = kRing - 1;
        if (closing && boundary) {
            const int kb = (t + 1) >> 1;
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                float nm;
                split_scale(sc[wpiece0[rt] * kMaxKBlocks + kb], pow2_next[rt], nm);
                ratio[rt] = uniform_f32(mant[rt] * __builtin_amdgcn_rcpf(nm));
                mant[rt] = nm;
            }
        }
        if (first) {   // conversions from here on belong to this K block (ratio == 1 and pow2_next == pow2 for t == 0)
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) pow2[rt] = pow2_next[rt];
        }
        kstep(0, true, false, false, dma_mid ? 2 : -1, dma_mid ? 3 : -1, t + 2, pbuf, false, first, false, chk);
        kstep(1, true, false, false, dma_mid ? 4 : -1, dma_mid ? 5 : -1, t + 2, pbuf, false, false, false, chk);
        kstep(2, true, false, false, -1, -1, 0, 0, false, false, false, chk);
        kstep(3, has_next, has_next, wait6, dma_tail ? 0 : -1, dma_tail ? 1 : -1, t + 3, buf, closing && boundary, false,
              closing && boundary, chk);
        buf = (buf + 1 == kRing) ? 0 : buf + 1;
    };
    // waves without rows (tail tiles): keep the DMA pieces and the sync points, skip the math
    auto idle_stage = [&](int t, bool has_next, bool wait6, bool dma_mid, bool dma_tail) __attribute__((always_inline)) {
        int pbuf = buf - 1;
        if (pbuf < 0) pbuf = kRing - 1;
        if (dma_mid) {
#pragma unroll
            for (int i = 2; i < 6; ++i) issue_piece(t + 2, pbuf, i);
        }
        if (has_next) sync_point(wait6);
        if (dma_tail) { issue_piece(t + 3, buf, 0); issue_piece(t + 3, buf, 1); }
        buf = (buf + 1 == kRing) ? 0 : buf + 1;
    };

    const bool active = wm * (NTT * 32) < rows;
#ifdef SGLK_DEV_ABLATE
    unsigned long long t0 = 0, r0 = 0;
    if (p.dbg && tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    // T = 2 * kblocks >= 4 stages.  Steady state while stage t+3 exists; the last two K blocks are peeled so that the
    // DMA / wait flags stay literals.
    // the main loop exists twice: for full tiles (every MFMA unconditional) and for tail tiles whose wave has fewer than NTT
    // token tiles with rows (a wave-uniform test in front of every token tile's read / MFMA / rescale)
    auto main_loop = [&](auto chk_c) __attribute__((always_inline)) {
        constexpr bool chk = decltype(chk_c)::value;
        // operands of (stage 0, k-step 0)
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) ld_w(rt, 0, 0);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) ld_x(0, tt, 0, 0, chk);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) { cvt2(0, rt, 0, pow2[rt]); cvt2(0, rt, 1, pow2[rt]); }
        SGLK_FENCE();
        int t = 0;
        for (; t + 4 < T; t += 2) {
            stage(t, true, false, true, true, true, true, true, chk);
            stage(t + 1, false, true, true, true, true, true, true, chk);
        }
        // t == T - 4 (when T >= 4): stages T-4 .. T-1
        if (t + 3 < T) {
            stage(t, true, false, true, true, true, true, true, chk);          // T-4: pieces 2..5 of T-2, pieces 0,1 of T-1
            stage(t + 1, false, true, true, true, true, false, true, chk);     // T-3: pieces 2..5 of T-1
            t += 2;
        }
        stage(t, true, false, true, false, false, false, true, chk);           // T-2: waits for all of T-1
        stage(t + 1, false, true, false, false, false, false, false, chk);     // T-1: nothing follows
    };
    if (active && (nta == NTT || !kTailSkip)) {
        main_loop(std::false_type{});
    } else if (active) {
        main_loop(std::true_type{});
    } else {
        int t = 0;
        for (; t + 4 < T; t += 2) {
            idle_stage(t, true, true, true, true);
            idle_stage(t + 1, true, true, true, true);
        }
        if (t + 3 < T) {
            idle_stage(t, true, true, true, true);
            idle_stage(t + 1, true, true, true, false);
            t += 2;
        }
        idle_stage(t, true, false, false, false);
        idle_stage(t + 1, false, false, false, false);
    }
#undef SGLK_FENCE
#ifdef SGLK_DEV_ABLATE
    if (p.dbg && tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        p.dbg[32 * L] = t1 - t0;
        p.dbg[32 * L + 1] = r1 - r0;
        p.dbg[32 * L + 18] = rt_entry;   // absolute 100 MHz ticks: entry, loop start, loop end
        p.dbg[32 * L + 19] = r0;
        p.dbg[32 * L + 20] = r1;
        p.dbg[32 * L + 22] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
        p.dbg[32 * L + 23] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
    }
#endif

    if (has_next_tile) fetch_meta(xs + jt_n, nxt);   // dependent row / scale loads of the next tile fly during the epilogue
    if (tid == 0) {
        // publish the tile after jt_nn: the ticket requested one tile ago, or the static stride (first tile, no counter)
        ticket_lds[0] = (p.tickets && my_ticket >= 0) ? 4 * nbx + my_ticket : jt_nn + nbx;
        if (p.tickets) my_ticket = atomicAdd(p.tickets + (blockIdx.x & 7), 1);
    }

    // ---- epilogue: accumulator -> LDS image [token][column] (16-B chunks XOR-swizzled by token&15) -> rows ------
    // 32x32 accumulator: lane = token column (l&31); register i = weight row (i&3) + 8*(i>>2) + 4*(l>>5) of the tile
    __syncthreads();   // every wave is done reading the ring
    // the epilogue's addressing is re-derived from an opaque copy of the thread id: hoisted out of the tile loop it
    // would sit in ~30 VGPRs across the main loop, which has none to spare
    int tidv = tid;
    asm volatile("" : "+v"(tidv));
    const int r32e = tidv & 31, he = (tidv >> 5) & 1;
#ifdef SGLK_DEV_ABLATE
    if (p.dbg && tid == 0) p.dbg[32 * L + 25] = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int kCols = (MODE == MODE_GATE_UP) ? 128 : 256;   // output columns per workgroup
    constexpr int kRowB = kCols * 2;                            // bytes per token row in the image
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        if (!active) break;
        const int r = wm * (NTT * 32) + tt * 32 + r32e;
        unsigned char* rowp = smem + r * kRowB;
        if (MODE == MODE_GATE_UP) {
#pragma unroll
            for (int gt = 0; gt < NRT / 2; ++gt) {   // gate row tile gt, up row tile gt + NRT/2: the same 32 ic1 columns
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    float g4[4], u4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        g4[i] = acc[gt][tt][rg * 4 + i] * mant[gt];
                        u4[i] = acc[gt + NRT / 2][tt][rg * 4 + i] * mant[gt + NRT / 2];
                    }
                    uint2 v;
                    v.x = pack_bf16x2(silu_f32(g4[0]) * u4[0], silu_f32(g4[1]) * u4[1]);
                    v.y = pack_bf16x2(silu_f32(g4[2]) * u4[2], silu_f32(g4[3]) * u4[3]);
                    const int col = wn * (NRT * 16) + gt * 32 + rg * 8 + he * 4;   // 4 consecutive ic1 columns
                    const int chunk = (col >> 3) ^ (r & 15);
                    *reinterpret_cast<uint2*>(rowp + chunk * 16 + (col & 4) * 2) = v;
                }
            }
        } else {
            float tw = 1.f;
            if (MODE == MODE_DOWN) tw = tw_tab[r];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                const float sc_w = mant[rt] * tw;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    float o4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) o4[i] = acc[rt][tt][rg * 4 + i] * sc_w;
                    const int col = wn * (NRT * 32) + rt * 32 + rg * 8 + he * 4;
                    if (MODE == MODE_PLAIN && p.ksplit > 1) {   // split-K: this range's fp32 partial sums, reduced by the caller
                        if (r < rows)
                            *reinterpret_cast<float4*>(p.partial + ((int64_t)e * p.split_rows + pos0 + r) * p.out_cols + ntile * kCols + col) =
                                make_float4(o4[0], o4[1], o4[2], o4[3]);
                        continue;
                    }
                    if (MODE == MODE_PLAIN) {   // dense epilogue: + bias[col] + addend[row][col] * scale, in fp32
                        const int gc = ntile * kCols + col;
                        if (p.bias) {
                            const float4 b = *reinterpret_cast<const float4*>(p.bias + gc);
                            o4[0] += b.x; o4[1] += b.y; o4[2] += b.z; o4[3] += b.w;
                        }
                        if (p.addend && r < rows) {
                            const uint2 a = *reinterpret_cast<const uint2*>(p.addend + (int64_t)(pos0 + r) * p.addend_stride + gc);
                            o4[0] += __uint_as_float(a.x << 16) * p.addend_scale;
                            o4[1] += __uint_as_float(a.x & 0xffff0000u) * p.addend_scale;
                            o4[2] += __uint_as_float(a.y << 16) * p.addend_scale;
                            o4[3] += __uint_as_float(a.y & 0xffff0000u) * p.addend_scale;
                        }
                    }
                    uint2 v;
                    v.x = pack_bf16x2(o4[0], o4[1]);
                    v.y = pack_bf16x2(o4[2], o4[3]);
                    const int chunk = (col >> 3) ^ (r & 15);
                    *reinterpret_cast<uint2*>(rowp + chunk * 16 + (col & 4) * 2) = v;
                }
            }
        }
    }
#ifdef SGLK_DEV_ABLATE
    if (p.dbg && tid == 0) p.dbg[32 * L + 26] = __builtin_amdgcn_s_memrealtime();
#endif
    __syncthreads();
#ifdef SGLK_DEV_ABLATE
    if (p.dbg && tid == 0) p.dbg[32 * L + 27] = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int kChunksPerRow = kRowB / 16;                   // 16 or 32
    constexpr int kIters = kBM * kChunksPerRow / 512;           // 8 or 16
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        const int idx = it * 512 + tidv;
        const int r = idx / kChunksPerRow;
        const int pc = idx - r * kChunksPerRow;                 // physical chunk
        const int lc = pc ^ (r & 15);                           // logical chunk = 8 columns
        if (r < rows && !(MODE == MODE_PLAIN && p.ksplit > 1)) {
            const uint4 v = *reinterpret_cast<const uint4*>(smem + r * kRowB + pc * 16);
            int64_t orow;
            if (MODE == MODE_DOWN) orow = (int64_t)slot_tab[r] * p.out_stride + ntile * kCols;
            else orow = (int64_t)(pos0 + r) * p.out_stride + ntile * kCols;
            *reinterpret_cast<uint4*>(p.out + orow + lc * 8) = v;
        }
    }
#ifdef SGLK_DEV_ABLATE
    if (p.dbg && tid == 0) {
        p.dbg[32 * L + 21] = __builtin_amdgcn_s_memrealtime();   // stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p.dbg[32 * L + 24] = __builtin_amdgcn_s_memrealtime();   // stores acknowledged
    }
#endif
    if (!has_next_tile) break;
    if (MODE == MODE_DOWN && nxt.my_slot >= 0) nxt.tw = p.topk_weights[nxt.my_slot];
    cur = nxt;
    jt = jt_n;
    jt_n = jt_nn;
    __syncthreads();   // the image and the row tables are dead: the next tile's DMA and tables may overwrite them
    jt_nn = __builtin_amdgcn_readfirstlane(ticket_lds[0]);
    }   // tile loop
}

}  // namespace g256i

// Persistent launch (one workgroup per CU walking a strided share of the tiles, next tile's tables prefetched) vs one
// workgroup per tile (hardware dispatch balances tail tiles better).  Returns the grid size; *persistent says which.
static int64_t plan_grid_256i(int C, int64_t tiles, bool* persistent) {
    const Knobs& kn = knobs();
    int cus = device_cu_count();
    const bool persist = kn.persist >= 0 ? kn.persist == 1 : C <= 1024;   // SGLK_PERSIST: A/B and test override
    if (persist && kn.max_wgs > 0 && kn.max_wgs < cus) cus = kn.max_wgs;   // SGLK_MAX_WGS: many tiles per workgroup on small problems
    const bool p = persist && tiles > cus;
    if (persistent) *persistent = p;
    return p ? cus : tiles;
}
bool moe_gemm_fp8w_256i_is_persistent(int C, int64_t tiles) {
    bool p = false;
    plan_grid_256i(C, tiles, &p);
    return p;
}

int launch_moe_gemm_fp8w_256i(int mode, const MoeGemmParams& p, int max_mtiles, hipStream_t stream) {
    int64_t blocks = (int64_t)max_mtiles * p.n_tiles;
    if (blocks == 0) return SGLK_OK;
    // Measured A/B on one box at the bench shape: persistent wins for the short reduction (K = 768: 0.464 vs 0.468 ms),
    // loses for the long one (K = 2048: 0.778 vs 0.765 ms) where the prologue is a smaller share and the static split's
    // imbalance costs more.
    const Knobs& kn = knobs();
    blocks = plan_grid_256i(p.C, blocks, nullptr);
    const bool wide_n = kn.wide_n;   // wave layout 2(n) x 4(m) instead of 4(n) x 2(m)
    if ((p.C >> 7) > g256i::kMaxKBlocks) SGLK_FAIL(SGLK_ERR_SHAPE, "moe_gemm_fp8w_256i: reduction length %d too long", p.C);
    if ((p.C >> 7) < 2 || p.C % 128 != 0) SGLK_FAIL(SGLK_ERR_SHAPE, "moe_gemm_fp8w_256i: reduction length %d needs at least two whole 128-wide K blocks", p.C);
    if (p.block_n % 32 != 0) SGLK_FAIL(SGLK_ERR_SHAPE, "moe_gemm_fp8w_256i: block_n %d is not a multiple of 32", p.block_n);
    if (mode == MODE_PLAIN) {
        if (wide_n) hipLaunchKernelGGL((g256i::moe_gemm_fp8w_256i_kernel<MODE_PLAIN, 0, 4>), dim3((unsigned)blocks), dim3(512), 0, stream, p);
        else hipLaunchKernelGGL((g256i::moe_gemm_fp8w_256i_kernel<MODE_PLAIN, 0, 2>), dim3((unsigned)blocks), dim3(512), 0, stream, p);
        SGLK_CHECK_LAUNCH("moe_gemm_fp8w_256i");
        return SGLK_OK;
    }
#define SGLK_LAUNCH256X(R)                                                                                             \
    if (mode == MODE_GATE_UP && wide_n)                                                                                \
        hipLaunchKernelGGL((g256i::moe_gemm_fp8w_256i_kernel<MODE_GATE_UP, R, 4>), dim3((unsigned)blocks), dim3(512), 0, stream, p); \
    else if (mode == MODE_GATE_UP)                                                                                     \
        hipLaunchKernelGGL((g256i::moe_gemm_fp8w_256i_kernel<MODE_GATE_UP, R, 2>), dim3((unsigned)blocks), dim3(512), 0, stream, p); \
    else if (wide_n)                                                                                                   \
        hipLaunchKernelGGL((g256i::moe_gemm_fp8w_256i_kernel<MODE_DOWN, R, 4>), dim3((unsigned)blocks), dim3(512), 0, stream, p); \
    else                                                                                                               \
        hipLaunchKernelGGL((g256i::moe_gemm_fp8w_256i_kernel<MODE_DOWN, R, 2>), dim3((unsigned)blocks), dim3(512), 0, stream, p)
#ifdef SGLK_DEV_ABLATE   // developer-only timing ablations (wrong results by design)
    const int abl = kn.rescale_ablate;
    switch (abl) {
        case 4: SGLK_LAUNCH256X(4); break;
        case 8: SGLK_LAUNCH256X(8); break;
        case 16: SGLK_LAUNCH256X(16); break;
        case 64: SGLK_LAUNCH256X(64); break;
        case 92: SGLK_LAUNCH256X(92); break;
        case 1024: SGLK_LAUNCH256X(1024); break;
        case 1116: SGLK_LAUNCH256X(1116); break;
        case 128: SGLK_LAUNCH256X(128); break;
        default: SGLK_LAUNCH256X(0); break;
    }
#else
    SGLK_LAUNCH256X(0);
#endif
#undef SGLK_LAUNCH256X
    SGLK_CHECK_LAUNCH("moe_gemm_fp8w_256i");
    return SGLK_OK;
}

}  // namespace sglk
