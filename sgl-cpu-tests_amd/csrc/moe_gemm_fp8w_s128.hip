// Large-M grouped W8A16 GEMM of fused_experts, two-term e4m3 split on the block-scaled fp8 matrix cores, 128-token tiles,
// TWO workgroups per CU.  Same operator contract, oracle and data formats as moe_gemm_fp8w_split.hip
// (/root/reference/test_moe_fp8_ext.py:22-25,70-91; /root/reference/bench_moe.py:113-130): x = hi + lo exactly, both terms e4m3
// under one power-of-two scale per (row, 128-wide block); rows are [hi 64 | lo 64] per 64-wide k group in the packed weight
// tile's k order.
//
// Why another tiling.  The 256 x 256 / 8-wave split kernel runs its main loop at 91 % of the matrix pipe but a tile also
// spends 3.6 us in its prologue (tile table -> row ids -> first operands: dependent round trips) and 7.2 us in its epilogue
// (SiLU, the split of ic1, stores), with the pipe idle: one workgroup owns the whole CU.  Here a workgroup is FOUR waves (one
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
//   * half the tile height of the 256-row kernels, and an expert's last tile (at most 96 rows; moe_align puts those into a second
//     table) skips the 32-token tiles that hold no row;
//   * PERSISTENT over the full tiles (per-XCD ticket counters): the next tile's table entry is fetched by a scalar load during
//     the main loop, its row ids, scales and its first two stages of activations + three stages of weights during the
//     EPILOGUE (ring slots 2, 3 and the weight-fragment registers are free then), so a workgroup goes from one main loop into
//     the next with no memory round trip in between -- the prologue was 4-5 us of every tile's 45 (GEMM-1) / 20 (GEMM-2).
#include <stddef.h>

#include <type_traits>

#include "fp8_split.h"
#include "knobs.h"
#include "moe_internal.h"

namespace sglk {

typedef __attribute__((address_space(3))) void* lptr_s1_t;
typedef __attribute__((ext_vector_type(8))) int i32x8;

namespace gs128 {

constexpr int kBM = 128;
constexpr int kStageX = kBM * 128;            // 16 KiB: 128 tokens x (hi 64 + lo 64) bytes
constexpr int kRing = 4;
constexpr int kMaxKB = 32;                    // reduction length <= 4096
constexpr int kImage = 64 * 1024;             // the ring (4 x 16 KiB) = the DOWN epilogue image (128 rows x 512 B)
constexpr int kScaleOff = kImage;                         // sc[16 pieces][kMaxKB] f32 (2 KiB)
constexpr int kXsOff = kScaleOff + 16 * kMaxKB * 4;       // xs[kb][128 tokens] E8M0 bytes (4 KiB)
constexpr int kRowTabOff = kXsOff + kMaxKB * kBM;         // DOWN: output slot + routing weight per tile row (1 KiB)
constexpr int kAmaxOff = kRowTabOff + 2 * kBM * 4;        // GATE_UP epilogue: amax[4 waves][128] f32 (2 KiB)
constexpr int kTicketOff = kAmaxOff + 4 * kBM * 4;        // the workgroup's tile ticket + the tile loop's bookkeeping (parked
                                                          // here across the main loop: the loop has no scalar register to spare)
constexpr int kParamOff = kTicketOff + 64;                // the kernel's parameter block, copied here once: the (non-inlined)
                                                          // tile functions read it from LDS instead of through a generic pointer
constexpr int kLds = kParamOff + 256;                     // 73.3 KiB: two workgroups per CU
static_assert(sizeof(A8GemmParams) <= 256 && sizeof(A8GemmParams) % 4 == 0, "parameter block");
constexpr int kSlot0 = 2;                                 // ring slot of a tile's stage 0: stage t lives in slot (t + 2) % 4, so that
                                                          // the NEXT tile's stages 0 and 1 can land in slots 2, 3 while the current
                                                          // tile's epilogue image occupies slots 0, 1

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

// pointers that went through LDS have lost their address space: every access through them says "global" again (a generic
// pointer makes flat_load / flat_store, which count in both wait counters)
template <class T>
SGLK_DEV __attribute__((address_space(1))) T* gp(T* q) {
    return (__attribute__((address_space(1))) T*)q;
}

// One LDS block for the kernel and the (non-inlined) tile functions it calls: every variant of a tile is its own function so
// that each gets its own register allocation (inlined into one kernel body they cost 200+ spills).
__shared__ __attribute__((aligned(16))) unsigned char smem[kLds];

// a field of the parameter block (LDS copy at kParamOff) as a scalar.  The reads are volatile: otherwise the compiler hoists
// them out of the tile loop (the block is never written there) and forty fields sit in scalar registers across the main loop,
// which has none to spare
template <class T>
SGLK_DEV T param_field(int off) {
    const volatile __attribute__((address_space(3))) int* w = (const volatile __attribute__((address_space(3))) int*)(smem + kParamOff + off);
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __builtin_amdgcn_readfirstlane(w[0]));
    } else {
        static_assert(sizeof(T) == 8, "4- or 8-byte fields");
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane(w[0]);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane(w[1]);
        return __builtin_bit_cast(T, ((unsigned long long)hi << 32) | lo);
    }
}
#define PF(f) param_field<decltype(A8GemmParams::f)>((int)offsetof(A8GemmParams, f))

struct TileId {
    int L, ntile, e, pos0, rows;   // L = linear work-item id (time stamps of developer builds only)
};

// Tiles of one workgroup.
// NMOD = (kblocks - 2) % 3: stage t multiplies with weight-fragment set (t + const) % 3, and the rotation is laid out from the
// END of the reduction so that every set index is a literal (a run-time phase switch around the K-block bodies cost the
// register allocator 1400 spills): K blocks in phases P0 (sets 0,1), P1 (2,0), P2 (1,2); the last two blocks are always P1, P2,
// the steady blocks before them whole (P1, P2, P0) triples preceded by a head of NMOD blocks (P0, or P2 P0).  (Four sets --
// weights three stages ahead -- measured the same and cost 16 registers the persistent loop does not have.)
// NTA = token tiles (32 tokens) of the tile that hold at least one row: 4 for the first tile table, 1..3 for an expert's short
// last tile (second table), which skips the fragment reads, MFMAs and rescales of the token tiles without rows.  The whole
// tile is compiled per count (a run-time test in front of every MFMA, or a switch around the main loop alone, cost the
// register allocator 30-100 spills).
// PERSIST: walk the XCD's share [xs, xs + xl) of the first table from index jt on (static second tile, then tickets), fetching
// the next tile inside the current one; else the one tile `first`.
template <int MODE, int NMOD, int ABL, int NTA, bool PERSIST>
SGLK_DEV void run_tiles(const TileId first_arg, const int xs_arg, const int xl_arg, const int nbx_arg, const int jt_arg) {
    // arguments of a non-inlined function arrive in vector registers: make the (wave-uniform) values scalar again
    auto uni = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
    TileId first;
    first.L = uni(first_arg.L);
    first.ntile = uni(first_arg.ntile);
    first.e = uni(first_arg.e);
    first.pos0 = uni(first_arg.pos0);
    first.rows = uni(first_arg.rows);
    const int xs_in = uni(xs_arg), xl_in = uni(xl_arg), nbx_in = uni(nbx_arg), jt_in = uni(jt_arg);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ctiles = PF(C) >> 6;      // 64-wide k groups = stages
    const int kblocks = PF(C) >> 7;

    float* sc = reinterpret_cast<float*>(smem + kScaleOff);          // sc[piece][kb]
    unsigned char* xs_tab = smem + kXsOff;                             // xs_tab[kb][token row]
    int* slot_tab = reinterpret_cast<int*>(smem + kRowTabOff);
    float* tw_tab = reinterpret_cast<float*>(smem + kRowTabOff + kBM * 4);
    float* amax_tab = reinterpret_cast<float*>(smem + kAmaxOff);
    int* ticket_lds = reinterpret_cast<int*>(smem + kTicketOff);   // [0] ticket, [4..11] loop bookkeeping

    // the workgroup's 16 packed 16-row weight pieces: GATE_UP = 8 gate + the 8 matching up pieces, DOWN = 16 consecutive
    auto piece_row16 = [&](int ntile, int piece) __attribute__((always_inline)) {
        if (MODE == MODE_GATE_UP) return (piece < 8) ? ntile * 8 + piece : (PF(n_half) >> 4) + ntile * 8 + (piece - 8);
        return ntile * 16 + piece;
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
    const int sw = (r32 >> 1) & 7;
    const int xo_h0 = r32 * 128 + (((2 * h) ^ sw) << 4), xo_h1 = r32 * 128 + (((2 * h + 1) ^ sw) << 4);
    const int xo_l0 = r32 * 128 + (((4 + 2 * h) ^ sw) << 4), xo_l1 = r32 * 128 + (((5 + 2 * h) ^ sw) << 4);

    // ---- what a tile needs besides its operands, fetched into registers (`Meta`) and written to the LDS tables later ----
    struct Meta {
        float sc_reg[2];               // scale-table entries sc[tid], sc[tid + 256]  (sc[piece * 32 + kb])
        unsigned xs_reg[kMaxKB / 4];   // tid < 128: the row's activation scale bytes
        int my_slot;                   // tid < 128: sorted_slot of tile row tid (-1 past the tile's rows)
        int slots[4];                  // GATE_UP: sorted_slot of the wave's four DMA row groups
    };
    auto fetch_slots = [&](const TileId& t, Meta& m) __attribute__((always_inline)) {
        m.my_slot = -1;
        if (tid < kBM && tid < t.rows) m.my_slot = gp(PF(sorted_slot))[t.pos0 + tid];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            m.slots[i] = 0;
            if (MODE == MODE_GATE_UP) {
                const int r = (wn * 4 + i) * 8 + (lane >> 3);
                m.slots[i] = gp(PF(sorted_slot))[t.pos0 + (r < t.rows ? r : 0)];
            }
        }
    };
    auto fetch_scales = [&](const TileId& t, Meta& m) __attribute__((always_inline)) {
        const float* scale_e = PF(w_scale) + (int64_t)t.e * PF(scale_rows) * PF(scale_cols);
        const float inv_bn = 1.0f / (float)PF(block_n);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * 256, piece = i >> 5, kb = i & (kMaxKB - 1);
            m.sc_reg[j] = 0.f;
            if (kb < kblocks) {
                // floor(row / block_n) through one float multiply (exact for rows < 2^20, see moe_gemm_fp8w_256i.hip)
                const int srow = (int)(((float)(piece_row16(t.ntile, piece) * 16) + 0.5f) * inv_bn);
                m.sc_reg[j] = gp(scale_e)[srow * PF(scale_cols) + kb];
            }
        }
    };
    auto fetch_xs = [&](const TileId& t, Meta& m) __attribute__((always_inline)) {   // after fetch_slots
#pragma unroll
        for (int i = 0; i < kMaxKB / 4; ++i) m.xs_reg[i] = 0x7f7f7f7fu;
        if (tid < kBM && tid < t.rows) {
            const int64_t xrow = (MODE == MODE_GATE_UP) ? (int64_t)(m.my_slot / PF(topk)) : (int64_t)(t.pos0 + tid);
            const unsigned* sp = reinterpret_cast<const unsigned*>(PF(xs) + xrow * PF(xs_stride));
#pragma unroll
            for (int i = 0; i < kMaxKB / 4; ++i)
                if (i * 4 < kblocks) m.xs_reg[i] = gp(sp)[i];
        }
    };
    // the main loop's tables (weight scales; activation scale bytes): both are dead once a main loop has ended, so the NEXT tile's
    // go in during the epilogue, as soon as their loads have landed -- parked in registers until the end they cost 10 of them
    auto store_tables = [&](const Meta& m, bool xs_part) __attribute__((always_inline)) {
        if (!xs_part) {
            sc[tid] = m.sc_reg[0];
            sc[tid + 256] = m.sc_reg[1];
        } else if (tid < kBM) {
#pragma unroll
            for (int i = 0; i < kMaxKB / 4; ++i)
                if (i * 4 < kblocks) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) xs_tab[(i * 4 + b) * kBM + tid] = (unsigned char)(m.xs_reg[i] >> (8 * b));
                }
        }
    };

    // ---- operand sources.  X: descriptor + one 32-bit lane offset per 1-KiB piece (8 rows x 128 B: lane = row l >> 3, chunk
    //      l & 7; image chunk = logical chunk ^ ((row >> 1) & 7), applied to the SOURCE address since an LDS-DMA lands
    //      lane-linear); the stage offset is the scalar soffset.  Rows past the tile's last one get an offset outside the
    //      descriptor's range and fetch nothing.  W: descriptor of the expert + one lane offset per 32-row tile. ----
    const unsigned xbytes = (unsigned)__builtin_amdgcn_readfirstlane((int)PF(x_bytes));
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)PF(x), 0, xbytes, 0x00020000);
    struct Src {
        unsigned x[4], w[2];
    };
    auto calc_src = [&](const TileId& t, const Meta& m, Src& s) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (wn * 4 + i) * 8 + (lane >> 3);
            unsigned off = xbytes;
            if (r < t.rows) {
                int64_t xrow;
                if (MODE == MODE_GATE_UP) xrow = (int64_t)(m.slots[i] / PF(topk));
                else xrow = (int64_t)(t.pos0 + r);
                off = (unsigned)(xrow * PF(x_stride)) + (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) << 4);
            }
            s.x[i] = off;
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
            s.w[rt] = (unsigned)(piece_row16(t.ntile, wpiece0[rt] + (r32 >> 4)) * ctiles) * 1024u + (unsigned)(((2 * h) * 16 + (r32 & 15)) * 16);
    };
    auto w_rsrc = [&](int e) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(PF(w) + (int64_t)e * PF(w_expert_stride)), 0, (unsigned)PF(w_expert_stride), 0x00020000);
    };
    auto issue_x = [&](const Src& s, int kt, int buf, int i) __attribute__((always_inline)) {   // piece i (0..3) of this wave, stage kt
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lptr_s1_t)(smem + buf * kStageX + (wn * 4 + i) * 1024), 16, s.x[i], kt * 128, 0, 0);
    };
    i32x8 fa[3][2] = {};                   // [(stage + const) % 3][row tile]
    auto ld_a = [&](const __amdgpu_buffer_rsrc_t rs, const Src& s, int as, int rt, int kt) __attribute__((always_inline)) {
        const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rs, s.w[rt], kt * 1024, 0);
        const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(rs, s.w[rt], kt * 1024 + 256, 0);   // + 16 slots: in the scalar offset
        fa[as][rt] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    };
    constexpr int kSet0 = NMOD == 0 ? 2 : (NMOD == 1 ? 0 : 1);   // fragment set of stage 0 = first set of K block 0's phase
    // first operands of a tile.  X(0), X(1): ring slots 2, 3 are free during an epilogue (the image uses slots 0, 1) and an LDS-DMA
    // needs no register, so they go out as early as the row ids are known.  A(0), A(1) pin 32 registers from request to use:
    // requested inside the epilogue they are carried across the tile loop's back edge, where the register allocator spilled
    // and reloaded them (the first MFMA of every tile then waited for vmcnt(0) behind a scratch reload).  They are requested
    // at the top of the tile's own iteration instead; the weights of an expert are shared by its m-tiles' workgroups, so this
    // is an L2 round trip, not an HBM one
    auto issue_head_x = [&](const Src& s) __attribute__((always_inline)) {
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_x(s, st, kSlot0 + st, i);
    };
    auto issue_head_a = [&](const __amdgpu_buffer_rsrc_t rs, const Src& s) __attribute__((always_inline)) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            ld_a(rs, s, (kSet0 + st) % 3, 0, st);
            ld_a(rs, s, (kSet0 + st) % 3, 1, st);
        }
    };
    // ... and X(2), half of X(3) into slots 0, 1 (the epilogue image's, once that is dead), the tables, and the wait for stage 0:
    // everything but these six pieces has landed (X(0), A(0), X(1), A(1) are older)
    auto issue_rest = [&](const Src& s, int my_slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_x(s, 2, 0, i);
        issue_x(s, 3, 1, 0);   // pieces 2, 3 of X(3) are carried into stage 0 like every later stage's
        issue_x(s, 3, 1, 1);
        if (MODE == MODE_DOWN && tid < kBM) slot_tab[tid] = my_slot;   // the finished tile's store pass has read its own
        __builtin_amdgcn_s_waitcnt(wc(6, 0));
        __builtin_amdgcn_s_barrier();
    };

    // ---- the first tile: the whole chain (row ids -> sources -> operands) ----
    TileId cur = first;
    Src src;
    __amdgpu_buffer_rsrc_t wrsrc = w_rsrc(cur.e);
    {
        Meta m;
        fetch_slots(cur, m);
        fetch_scales(cur, m);
        fetch_xs(cur, m);
        calc_src(cur, m, src);
        issue_head_x(src);
        store_tables(m, false);
        store_tables(m, true);
        issue_rest(src, m.my_slot);
    }

    // tile sequence (PERSIST): the second tile is static (jt + nbx), later ones come from the XCD's ticket counter: a ticket
    // is requested at the start of an epilogue and becomes the tile after next at its end
    int jt_n = jt_in + nbx_in;

    for (;;) {   // ---- one tile per iteration -------------------------------------------------------------------------------
#ifdef SGLK_DEV_ABLATE
    const int L = cur.L;
    const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
#define SGLK_STAMP(i) do { if (PF(dbg) && tid == 0) gp(PF(dbg))[32 * L + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SGLK_STAMP(i) do { } while (0)
#endif
    issue_head_a(wrsrc, src);
    if (tid == 0) {   // parked in LDS across the main loop (read back behind the barrier that opens the epilogue)
        ticket_lds[4] = jt_n;
        ticket_lds[5] = cur.ntile;
        ticket_lds[6] = cur.pos0;
        ticket_lds[7] = cur.rows;
        ticket_lds[8] = cur.L;
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][tt][i] = 0.f;
    float my_tw = 0.f;

    int ea[2], ea_next[2];
    float mant[2], ratio[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        split_scale(sc[wpiece0[rt] * kMaxKB], ea[rt], mant[rt]);
        ea_next[rt] = ea[rt];
        ratio[rt] = 1.f;
    }
    // B scale bytes of the lane's four tokens for the current K block (the lo term's scale is this - 4); token tile tt's byte
    // of the NEXT block is read into the same register right behind the tile's last MFMA of the block
    int xsv[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) xsv[tt] = xs_tab[tt * 32 + r32];

#define SGLK_FENCE() __builtin_amdgcn_sched_barrier(0)
    // ONE window of token fragments (hi / lo of a 32-token tile): the next tile's fragment is requested right behind the last
    // MFMA that reads the current one, two to three MFMAs (128+ cycles) ahead of its first use.  (A second window, requests
    // four MFMAs ahead, measured the same: 0.6637 vs 0.6600 ms.)
    // ABL (developer builds, wrong results by design): 1 = no accumulator rescale, 2 = no activation DMA in the steady state,
    // 4 = no weight loads in the steady state, 8 = no barrier
    i32x8 bh = {}, bl = {};
    float nsc[2] = {0.f, 0.f};
    auto ld_bh = [&](int tt, int buf) __attribute__((always_inline)) {
        const i32x4 a0 = *reinterpret_cast<const i32x4*>(smem + (buf * kStageX + xo_h0) + tt * 4096);
        const i32x4 a1 = *reinterpret_cast<const i32x4*>(smem + (buf * kStageX + xo_h1) + tt * 4096);
        bh = i32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    };
    auto ld_bl = [&](int tt, int buf) __attribute__((always_inline)) {
        const i32x4 a0 = *reinterpret_cast<const i32x4*>(smem + (buf * kStageX + xo_l0) + tt * 4096);
        const i32x4 a1 = *reinterpret_cast<const i32x4*>(smem + (buf * kStageX + xo_l1) + tt * 4096);
        bl = i32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    };
    // MFMA slot s of a stage (16 per wave): token tile s >> 2, then hi x rt0, hi x rt1, lo x rt0, lo x rt1
    auto mma = [&](int as, int s2) __attribute__((always_inline)) {
        const int tt = s2 >> 2, lo = (s2 >> 1) & 1, rt = s2 & 1;
        if (lo)
            acc[rt][tt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[as][rt], bl, acc[rt][tt], 0, 0, 0, ea[rt], 0, xsv[tt] - 4);
        else
            acc[rt][tt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[as][rt], bh, acc[rt][tt], 0, 0, 0, ea[rt], 0, xsv[tt]);
    };
    // accumulator tile (rt, tt) into units of the next K block's mantissa, four registers (chunk c) at a time: a lone wave is
    // issue-bound, so the 128 multiplies of a K-block boundary are spread four per tile and MFMA slot
    auto rescale4 = [&](int rt, int tt, int c) __attribute__((always_inline)) {
        if (ABL & 1) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) asm("v_mul_f32 %0, %1, %0" : "+v"(acc[rt][tt][c * 4 + i]) : "s"(ratio[rt]));
    };
    // chunks due behind MFMA slot g (0 .. 31 over the closing stage and the first stage of the next block): tile (rt, tt) is
    // free from slot 4 tt + 3 + rt of the closing stage (its last MFMA of the block was slot 4 tt + 2 + rt) until its first
    // MFMA of the next block, slot 16 + 4 tt + rt; chunk c goes behind slot 4 tt + 3 + rt + c
    auto rescale_slot = [&](int g) __attribute__((always_inline)) {
#pragma unroll
        for (int tt = 0; tt < NTA; ++tt)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int c = g - (4 * tt + 3 + rt);
                if (c >= 0 && c < 4) rescale4(rt, tt, c);
            }
    };

    int buf = kSlot0;
    SGLK_STAMP(19);
#ifdef SGLK_DEV_ABLATE
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
#endif
    // Stage t (ring slot `buf`, weight fragments fa[as]); every flag is a literal at the call site.
    //   first  : first stage of a K block -- the weight scale switches, the rest of the boundary's rescale chunks
    //   pre    : first stage of a K block that is not the last: the NEXT block's weight scales are read from the LDS table
    //   bound  : closing stage of a K block that is not the last -- rescale chunks, and the next block's activation scale bytes
    //            are read (behind each token tile's last MFMA)
    //   lda    : stage t+2 exists: its weight fragments are requested (slots 0, 4) into fa[(as + 2) % 3], dead since stage t-1
    //   wait   : >= 0: stage t+1 exists; sync point after slot 13 = s_waitcnt vmcnt(wait) (X(t+1) and A(t+1) have landed; the nine
    //            operations issued behind them -- the last piece of X(t+2), half of X(t+3), A(t+2), the rest of X(t+3) -- may
    //            stay in flight; the literals 9 / 5 / 0 come from replaying the issue order, tools/s128_waits.py) + lgkmcnt(0)
    //            + barrier.  Every fragment of THIS stage has been read by then, so afterwards X(t+4) goes into this stage's
    //            ring slot and the first fragments of stage t+1 are read
    //   dmax   : stage t+4 exists: pieces 0, 1 of X(t+4) go out behind slots 14, 15 (one LDS-DMA per slot: a piece costs the
    //            issuing wave 60-180 cycles, more than one MFMA's shadow), pieces 2, 3 are carried into stage t+1
    //   carry  : pieces 2, 3 of X(t+3) (slots 2, 6), whose pieces 0, 1 the previous stage issued
    auto stage = [&](int as, int t, bool first, bool bound, int wait, bool dmax, bool lda, bool pre, bool carry) __attribute__((always_inline)) {
        int nbuf = buf + 1;
        if (nbuf == kRing) nbuf = 0;
        const int as2 = (as + 2) % 3;
        if (first) {   // this block's weight scale (computed one stage ago) becomes current
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) ea[rt] = ea_next[rt];
        }
        if (pre) {
            const int kb = (t >> 1) + 1;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) nsc[rt] = sc[wpiece0[rt] * kMaxKB + kb];
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
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const int tt = s2 >> 2, q = s2 & 3;
            if (tt < NTA) mma(as, s2);
            SGLK_FENCE();
            // the hi fragment is free after the tile's second MFMA, the lo fragment after its fourth
            if (q == 1 && tt + 1 < NTA) ld_bh(tt + 1, buf);
            if (q == 3 && tt + 1 < NTA) ld_bl(tt + 1, buf);
            if (q == 3 && bound && tt < NTA) xsv[tt] = xs_tab[((t >> 1) + 1) * kBM + tt * 32 + r32];
            if (s2 == 0 && lda && !(ABL & 4)) ld_a(wrsrc, src, as2, 0, t + 2);
            if (s2 == 4 && lda && !(ABL & 4)) ld_a(wrsrc, src, as2, 1, t + 2);
            if (s2 == 2 && carry && !(ABL & 2)) issue_x(src, t + 3, buf == 0 ? kRing - 1 : buf - 1, 2);
            if (s2 == 6 && carry && !(ABL & 2)) issue_x(src, t + 3, buf == 0 ? kRing - 1 : buf - 1, 3);
            if (bound) rescale_slot(s2);
            if (first) rescale_slot(16 + s2);
            if (s2 == 13 && wait >= 0) {
                if (wait == 9) __builtin_amdgcn_s_waitcnt(wc(9, 0));
                else if (wait == 5) __builtin_amdgcn_s_waitcnt(wc(5, 0));
                else __builtin_amdgcn_s_waitcnt(wc(0, 0));
                if (!(ABL & 8)) __builtin_amdgcn_s_barrier();
                ld_bh(0, nbuf);
            }
            if (s2 == 14 && dmax && !(ABL & 2)) issue_x(src, t + 4, buf, 0);
            if (s2 == 15 && wait >= 0) ld_bl(0, nbuf);
            if (s2 == 15 && dmax && !(ABL & 2)) issue_x(src, t + 4, buf, 1);
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
            stage(a0, t, true, false, 9, true, true, true, true);
            stage(a1, t + 1, false, true, 9, true, true, false, true);
        } else if (kind == KB_PENULT) {
            stage(a0, t, true, false, 9, false, true, true, true);
            stage(a1, t + 1, false, true, 5, false, true, false, false);
        } else {
            if (MODE == MODE_DOWN && tid < kBM) {   // the row's routing weight: covered by stage T-2's vmcnt(0)
                const int slot = slot_tab[tid];
                if (slot >= 0) my_tw = gp(PF(topk_weights))[slot];
            }
            stage(a0, t, true, false, 0, false, false, false, false);
            stage(a1, t + 1, false, false, -1, false, false, false, false);
        }
    };
    {
        ld_bh(0, kSlot0);
        ld_bl(0, kSlot0);
        __builtin_amdgcn_s_waitcnt(wc(0, 15));   // A(0), A(1) (and whatever the epilogue left in flight: X(2), X(3) halves)
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
#ifdef SGLK_DEV_ABLATE
    if (PF(dbg) && tid == 0) {   // shader clocks / 100 MHz ticks over the main loop -> the clock the chip held
        gp(PF(dbg))[32 * L + 0] = __builtin_amdgcn_s_memtime() - clk0;
        gp(PF(dbg))[32 * L + 1] = gp(PF(dbg))[32 * L + 20] - gp(PF(dbg))[32 * L + 19];
    }
#endif
    if (MODE == MODE_DOWN && tid < kBM) tw_tab[tid] = my_tw;

    // ---- epilogue (ring dead).  32x32 accumulator: lane = token column (l & 31); register i = weight row
    //      (i & 3) + 8 * (i >> 2) + 4 * (l >> 5) of the row tile.  The next tile's row ids, scales, first stages of activations
    //      (ring slots 2, 3; the image uses slots 0, 1) and three stages of weights are requested along the way ----
    __syncthreads();
    SGLK_STAMP(25);
    // the parameter block is re-read from here on (an opaque copy of its address): what the main loop keeps of it in scalar
    // registers is not carried across
    TileId ct;   // the tile being finished
    ct.ntile = __builtin_amdgcn_readfirstlane(ticket_lds[5]);
    ct.pos0 = __builtin_amdgcn_readfirstlane(ticket_lds[6]);
    ct.rows = __builtin_amdgcn_readfirstlane(ticket_lds[7]);
    ct.L = __builtin_amdgcn_readfirstlane(ticket_lds[8]);
    ct.e = 0;
    const int jt_next = PERSIST ? __builtin_amdgcn_readfirstlane(ticket_lds[4]) : 0;
    const bool has_next = PERSIST && jt_next < xl_in;
    int my_ticket = -1;
    if (has_next && tid == 0 && PF(tickets)) my_ticket = __hip_atomic_fetch_add(gp(PF(tickets)) + (blockIdx.x & 7), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int tidv = tid;
    asm volatile("" : "+v"(tidv));
    const int r32e = tidv & 31, he = (tidv >> 5) & 1;
    TileId nxt = ct;
    Meta nm;
    Src nsrc;
    __amdgpu_buffer_rsrc_t nrsrc = wrsrc;
    if (has_next) {
        nxt.L = xs_in + jt_next;
        const int mt = nxt.L / PF(n_tiles);
        nxt.ntile = nxt.L - mt * PF(n_tiles);
        const i32x4 ti = *gp(reinterpret_cast<const i32x4*>(PF(tile_info) + mt));
        nxt.e = __builtin_amdgcn_readfirstlane(ti[0]);
        nxt.pos0 = __builtin_amdgcn_readfirstlane(ti[1]);
        nxt.rows = __builtin_amdgcn_readfirstlane(ti[2]);
        fetch_slots(nxt, nm);
        fetch_scales(nxt, nm);
        if (MODE == MODE_DOWN) {   // positions, not row ids: everything can go out at once
            fetch_xs(nxt, nm);
            calc_src(nxt, nm, nsrc);
            nrsrc = w_rsrc(nxt.e);
            issue_head_x(nsrc);
        }
    }
    if (MODE == MODE_GATE_UP) {
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
                const unsigned pk = pack_bf16x2(silu_f32(g0) * u0, silu_f32(g1) * u1);
                vp[tt][i >> 1] = pk;
                am = fmaxf(am, fmaxf(fabsf(__uint_as_float(pk << 16)), fabsf(__uint_as_float(pk & 0xffff0000u))));
            }
            am = fmaxf(am, __shfl_xor(am, 32));
            if (he == 0) amax_tab[wn * kBM + tt * 32 + r32e] = am;
        }
        if (has_next) {   // the row ids have had the SiLU pass to arrive
            store_tables(nm, false);
            fetch_xs(nxt, nm);
            calc_src(nxt, nm, nsrc);
            nrsrc = w_rsrc(nxt.e);
            issue_head_x(nsrc);
        }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int r = tt * 32 + r32e;
            const float am = fmaxf(fmaxf(amax_tab[r], amax_tab[kBM + r]), fmaxf(amax_tab[2 * kBM + r], amax_tab[3 * kBM + r]));
            const int sb = sp_e8m0_for_amax(am);
            if (wn == 0 && he == 0 && r < ct.rows) gp(PF(out_s))[(int64_t)(ct.pos0 + r) * PF(out_s_stride) + ct.ntile] = (uint8_t)sb;
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
        if (has_next) store_tables(nm, true);   // the activation scale bytes requested before the amax exchange
        SGLK_STAMP(26);
        __syncthreads();
        SGLK_STAMP(27);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 256 + tidv;
            const int r = idx >> 4, pc = idx & 15, lc = pc ^ (r & 15);
            if (r < ct.rows) {
                const u32x4 val = *reinterpret_cast<const u32x4*>(smem + r * 256 + pc * 16);
                *gp(reinterpret_cast<u32x4*>((unsigned char*)PF(out) + (int64_t)(ct.pos0 + r) * PF(out_stride) + ct.ntile * 256 + lc * 16)) = val;
            }
        }
    } else {
        // ic2[slot] = topk_w * (acc * mant) in bf16: XOR-swizzled [token][256 columns] image in two halves of 64 tokens (32 KiB:
        // ring slots 0, 1), whole rows out by slot
        constexpr int kRowB = 512;
        uint16_t* outp = reinterpret_cast<uint16_t*>(PF(out));
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half == 1) __syncthreads();   // the first half's rows have been read
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int tt = half * 2 + t2;
                const int r = tt * 32 + r32e;
                unsigned char* rowp = smem + (r - half * 64) * kRowB;
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
            if (half == 0) SGLK_STAMP(26);
            if (half == 0 && has_next) {   // the next tile's tables: their loads went out at the start of the epilogue
                store_tables(nm, false);
                store_tables(nm, true);
            }
            __syncthreads();
            if (half == 0) SGLK_STAMP(27);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int idx = it * 256 + tidv;
                const int rl = idx >> 5, pc = idx & 31;
                const int r = half * 64 + rl, lc = pc ^ (r & 15);
                if (r < ct.rows) {
                    const u32x4 val = *reinterpret_cast<const u32x4*>(smem + rl * kRowB + pc * 16);
                    *gp(reinterpret_cast<u32x4*>(outp + (int64_t)slot_tab[r] * PF(out_stride) + ct.ntile * 256 + lc * 8)) = val;
                }
            }
        }
    }
#ifdef SGLK_DEV_ABLATE
    if (PF(dbg) && tid == 0) {
        gp(PF(dbg))[32 * ct.L + 18] = rt_entry;
        gp(PF(dbg))[32 * ct.L + 22] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
        gp(PF(dbg))[32 * ct.L + 23] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
        gp(PF(dbg))[32 * ct.L + 21] = __builtin_amdgcn_s_memrealtime();   // stores issued
        if (!has_next) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            gp(PF(dbg))[32 * ct.L + 24] = __builtin_amdgcn_s_memrealtime();   // stores acknowledged
        } else {
            gp(PF(dbg))[32 * ct.L + 24] = gp(PF(dbg))[32 * ct.L + 21];
        }
    }
#endif
#undef SGLK_STAMP
    if (!has_next) break;
    if (tid == 0) ticket_lds[0] = my_ticket >= 0 ? 2 * nbx_in + my_ticket : jt_next + nbx_in;   // the tile after next
    __syncthreads();   // the image and the row tables are dead: the next tile's pieces and tables may overwrite them
    jt_n = __builtin_amdgcn_readfirstlane(ticket_lds[0]);
    issue_rest(nsrc, nm.my_slot);
    cur = nxt;
    src = nsrc;
    wrsrc = nrsrc;
    }   // tile loop
}

// a tile variant as a function of its own (own register allocation): the persistent form calls these
template <int MODE, int NMOD, int ABL, int NTA, bool PERSIST>
__device__ __attribute__((noinline)) void run_tiles_fn(const TileId first, const int xs, const int xl, const int nbx, const int jt) {
    run_tiles<MODE, NMOD, ABL, NTA, PERSIST>(first, xs, xl, nbx, jt);
}

SGLK_DEV TileId tile_from_table(const int4* table, int item, int n_tiles, int L) {
    TileId t;
    t.L = L;
    const int mt = item / n_tiles;
    t.ntile = item - mt * n_tiles;
    const int4 ti = table[mt];
    t.e = __builtin_amdgcn_readfirstlane(ti.x);
    t.pos0 = __builtin_amdgcn_readfirstlane(ti.y);
    t.rows = __builtin_amdgcn_readfirstlane(ti.z);
    return t;
}

// PERSISTK = false (what ships): one workgroup per tile of the (single) tile table, dispatched by the hardware; the tile's code is
// chosen by its count of token tiles with rows.  PERSISTK = true (SGLK_S128_PERSIST=1, experimental): two resident workgroups per
// CU walk the table of tiles with more than 96 rows by ticket, fetching the next tile inside the epilogue, then the experts'
// short last tiles (second table).  On ROCm 7.2 the register allocator spills ~170 values per tile in that form (the tile loop
// is at the limit of both register files), which costs more than the hidden prologue returns: 1.13 vs 0.63 ms for GEMM-1.
template <int MODE, int NMOD, int ABL, bool PERSISTK>
__global__ __launch_bounds__(256, 2) void moe_gemm_fp8w_s128_kernel(const A8GemmParams p) {
#ifdef __HIP_DEVICE_COMPILE__   // hipcc's HOST pass (ROCm 7.2) rejects the calls of the tail-tile variants below ("candidate template
                                // ignored: substitution failure", no reason given; the device pass takes them): it only needs the stub
    if (threadIdx.x < sizeof(A8GemmParams) / 4)
        reinterpret_cast<int*>(smem + kParamOff)[threadIdx.x] = reinterpret_cast<const int*>(&p)[threadIdx.x];
    __syncthreads();
    // XCD x owns the contiguous range [xs, xs + xl) of (m-tile, column tile) pairs, column tiles fastest: the column tiles of an
    // m-tile share its gathered rows, the m-tiles of an expert its weights, both inside one L2
    const int live = p.num_tiles[0] * p.n_tiles;
    const int x = blockIdx.x & 7, q = live >> 3, r = live & 7;
    const int xs = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    const int xl = q + (x < r ? 1 : 0);
    const int jt = blockIdx.x >> 3;
    if constexpr (!PERSISTK) {
        if (jt >= xl) return;
        const TileId t = tile_from_table(p.tile_info, xs + jt, p.n_tiles, xs + jt);
        const int nta = (t.rows + 31) >> 5;
        if (nta >= 4) run_tiles<MODE, NMOD, ABL, 4, false>(t, 0, 0, 0, 0);
        else if (nta == 3) run_tiles<MODE, NMOD, ABL, 3, false>(t, 0, 0, 0, 0);
        else if (nta == 2) run_tiles<MODE, NMOD, ABL, 2, false>(t, 0, 0, 0, 0);
        else run_tiles<MODE, NMOD, ABL, 1, false>(t, 0, 0, 0, 0);
    } else {
        // ---- phase 1: the tiles of more than 96 rows, persistent ----
        if (jt < xl) {
            const int nbx = ((int)gridDim.x - x + 7) >> 3;   // workgroups of this launch on XCD x
            const TileId t = tile_from_table(p.tile_info, xs + jt, p.n_tiles, xs + jt);
            run_tiles_fn<MODE, NMOD, ABL, 4, true>(t, xs, xl, nbx, jt);
        }
        // ---- phase 2: the experts' short last tiles (second table), one at a time by ticket ----
        if (!p.tile_info_b) return;
        const int nb = p.num_tiles_b[0] * p.n_tiles;
        int* ticket_lds = reinterpret_cast<int*>(smem + kTicketOff);
        for (;;) {
            __syncthreads();   // the previous tile's image and tables are dead
            if (threadIdx.x == 0) ticket_lds[0] = atomicAdd(p.tickets + 8, 1);
            __syncthreads();
            const int it = __builtin_amdgcn_readfirstlane(ticket_lds[0]);
            if (it >= nb) break;
            const TileId t = tile_from_table(p.tile_info_b, it, p.n_tiles, live + it);
            const int nta = (t.rows + 31) >> 5;
            if (nta >= 3) run_tiles_fn<MODE, NMOD, ABL, 3, false>(t, 0, 0, 0, 0);
            else if (nta == 2) run_tiles_fn<MODE, NMOD, ABL, 2, false>(t, 0, 0, 0, 0);
            else run_tiles_fn<MODE, NMOD, ABL, 1, false>(t, 0, 0, 0, 0);
        }
    }
#endif
}

}  // namespace gs128

bool moe_gemm_fp8w_s128_ok(int N, int K, int block_n) {
    // both reductions (K for GEMM-1, N for GEMM-2) in whole 128-wide blocks, 2 .. 32 of them; 128 ic1 columns / 256 output
    // columns per workgroup; a 32-row operand tile inside one scale block
    return K % 256 == 0 && N % 128 == 0 && K >= 256 && N >= 256 && K <= 128 * gs128::kMaxKB && N <= 128 * gs128::kMaxKB &&
           block_n % 32 == 0;
}

bool moe_gemm_fp8w_s128_persistent() { return knobs().s128_persist == 1; }

int launch_moe_gemm_fp8w_s128(int mode, const A8GemmParams& p, int max_mtiles, hipStream_t stream) {
    int64_t blocks = (int64_t)max_mtiles * p.n_tiles;
    if (blocks == 0) return SGLK_OK;
    const bool persist = moe_gemm_fp8w_s128_persistent();
    if (persist) {
        // two workgroups per CU walk the tile tables (fewer when there are fewer tiles)
        const int64_t resident = 2 * (int64_t)device_cu_count();
        if (blocks > resident) blocks = resident;
        if (!p.tickets || !p.tile_info_b || !p.num_tiles_b) SGLK_FAIL(SGLK_ERR_INVALID, "moe_gemm_fp8w_s128: tickets / second tile table missing");
    }
    blocks = (blocks + 7) / 8 * 8;   // every XCD's share of the tile list must be reachable (blockIdx >> 3)
    const int kblocks = p.C >> 7;
    if (p.C % 128 != 0 || kblocks < 2 || kblocks > gs128::kMaxKB)
        SGLK_FAIL(SGLK_ERR_SHAPE, "moe_gemm_fp8w_s128: reduction length %d (needs 2..%d whole 128-wide K blocks)", p.C, gs128::kMaxKB);
    if (p.block_n % 32 != 0) SGLK_FAIL(SGLK_ERR_SHAPE, "moe_gemm_fp8w_s128: block_n %d is not a multiple of 32", p.block_n);
    if (p.xs_stride % 4 != 0 || ((uintptr_t)p.xs % 4) != 0) SGLK_FAIL(SGLK_ERR_INVALID, "moe_gemm_fp8w_s128: scale rows must be 4-byte aligned");
    if (mode != MODE_GATE_UP && mode != MODE_DOWN) SGLK_FAIL(SGLK_ERR_INVALID, "moe_gemm_fp8w_s128: mode %d", mode);
    const int nmod = (kblocks - 2) % 3;
#define SGLK_LAUNCH_S128(M_, N_, A_, P_) \
    hipLaunchKernelGGL((gs128::moe_gemm_fp8w_s128_kernel<M_, N_, A_, P_>), dim3((unsigned)blocks), dim3(256), 0, stream, p)
#ifdef SGLK_DEV_ABLATE   // developer-only timing ablations (wrong results by design)
    const int abl = knobs().rescale_ablate;
#define SGLK_LAUNCH_S128_B(M_, N_)                                                                       \
    do {                                                                                                 \
        if (persist) SGLK_LAUNCH_S128(M_, N_, 0, true);                                                  \
        else switch (abl) {                                                                              \
            case 1: SGLK_LAUNCH_S128(M_, N_, 1, false); break;                                           \
            case 2: SGLK_LAUNCH_S128(M_, N_, 2, false); break;                                           \
            case 4: SGLK_LAUNCH_S128(M_, N_, 4, false); break;                                           \
            case 7: SGLK_LAUNCH_S128(M_, N_, 7, false); break;                                           \
            default: SGLK_LAUNCH_S128(M_, N_, 0, false); break;                                          \
        }                                                                                                \
    } while (0)
#else
#define SGLK_LAUNCH_S128_B(M_, N_) do { if (persist) SGLK_LAUNCH_S128(M_, N_, 0, true); else SGLK_LAUNCH_S128(M_, N_, 0, false); } while (0)
#endif
    if (mode == MODE_GATE_UP) {
        if (nmod == 0) SGLK_LAUNCH_S128_B(MODE_GATE_UP, 0);
        else if (nmod == 1) SGLK_LAUNCH_S128_B(MODE_GATE_UP, 1);
        else SGLK_LAUNCH_S128_B(MODE_GATE_UP, 2);
    } else {
        if (nmod == 0) SGLK_LAUNCH_S128_B(MODE_DOWN, 0);
        else if (nmod == 1) SGLK_LAUNCH_S128_B(MODE_DOWN, 1);
        else SGLK_LAUNCH_S128_B(MODE_DOWN, 2);
    }
#undef SGLK_LAUNCH_S128_B
#undef SGLK_LAUNCH_S128
    SGLK_CHECK_LAUNCH("moe_gemm_fp8w_s128");
    return SGLK_OK;
}

}  // namespace sglk
