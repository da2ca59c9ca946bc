// Mid-M grouped W8A16 GEMM of fused_experts: weight-STREAMING kernel for experts that see a few dozen to ~128 rows
// (M ~ 300 .. 2000 at Qwen3-30B-A3B).  The call is still bound by reading every expert's fp8 weights once (604 MB),
// but a 32-token tile (moe_gemm_fp8w_stream.hip) would read them two to four times and a 256-token tile
// (moe_gemm_fp8w_256i.hip) spends its LDS ring and half its waves on padding rows.
//
// Same math contract as moe_gemm_fp8w.hip (oracle /root/reference/test_moe_fp8_ext.py:22-25,70-91).
//
// * tile = up to 96 tokens x 16 (GATE_UP) or 8 (DOWN) weight row-tiles per workgroup of 8 waves; every wave owns two
//   16-row weight tiles (GATE_UP: the gate tile and the matching up tile, so SiLU*mul stays in-register) or one (DOWN:
//   two workgroups per CU) for ALL the tile's tokens;
// * the number of 16-token MFMA column tiles is the tile's own (MT = ceil(rows / 16), rounded to 2 / 4 / 6 and
//   compiled as separate loop bodies): math, LDS traffic and accumulator registers follow the rows that exist;
// * weights never touch LDS: a packed piece (16 rows x 64 k = 1 KiB, pack.hip) is one global_load_dwordx4 per lane and
//   lands in MFMA A-operand order.  Each wave keeps 8 pieces = two 128-wide K blocks of its two tiles in flight
//   (statically indexed registers), 64 KiB per CU;
// * activations go through LDS one 128-wide K block at a time (tokens x 256 B, 16-byte chunks XOR-swizzled by
//   row & 15), double buffered: block kb+1 travels global -> LDS by DMA (no registers; the swizzle is applied to the
//   source address) while block kb is multiplied, one counted s_waitcnt + one barrier per block.  Each token fragment read feeds both of the wave's
//   weight tiles (8 MFMAs per 4 ds_read_b128);
// * fp8 -> bf16 exactly (v_cvt_scalef32_pk_bf16_fp8, scale 1.0); the 128-block partial sum lives in a temporary and
//   is folded into the accumulator with the block scale in fp32.
#include "knobs.h"
#include "moe_internal.h"

namespace sglk {
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

namespace gmid {

constexpr int kTM = kMidTileM;        // 128 tokens
constexpr int kXBuf = kTM * 256;      // one K block of the token tile: 32 KiB
constexpr int kMaxKB = 64;             // K blocks a workgroup may walk (reduction length or split-K range <= 8192)
constexpr int kScOff = 2 * kXBuf;     // then the scale table [8 waves][2 tiles][kMaxKB] f32
constexpr int kRowTabOff = kScOff + 8 * 2 * kMaxKB * 4;   // DOWN: output slot and routing weight of every tile row
constexpr int kLds = kRowTabOff + 2 * kTM * 4;

SGLK_DEV bf16x8 cvt8(unsigned lo, unsigned hi) {
    const bf16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false);
    const bf16x2 b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
    const bf16x2 c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false);
    const bf16x2 d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1];
    r[4] = c[0]; r[5] = c[1]; r[6] = d[0]; r[7] = d[1];
    return r;
}

struct TileCtx {
    int pos0, rows, ntile, kblocks;   // kblocks = 128-wide K blocks THIS workgroup walks (the whole reduction, or its range)
    int kb0, ksr;                     // PLAIN split-K: first block and index of the range
    const unsigned char* wp[2];   // lane's byte inside the first piece of the wave's two weight row-tiles
    const float* sc;              // LDS: this wave's [2][32] block scales
    int row16[2];
};

// Everything after the tile lookup, for MT 16-token column tiles.  TPW = weight row-tiles per wave: 2 for GATE_UP (gate +
// up), 1 for DOWN -- half the registers, so that two workgroups share a CU and one's prologue (tile table -> rows ->
// first activations, three dependent round trips) and epilogue hide behind the other's stream; DOWN's reduction is
// short (N = 768: six K blocks), so without that overlap the prologue is a third of its time.
// NW = waves per workgroup = 16-row weight tiles (GATE_UP: tile pairs) it covers: 8, or 4 / 2 for GATE_UP launches that would leave
// most of the chip idle (decode-size batches: a CU streams ~36 GB/s whatever its ring depth, so what counts is how many CUs the
// launch reaches -- the reference's decode shape, bench_moe.py:144, M = 4 / E = 256 / N = 384, is 96 workgroups of eight waves)
// XD = K blocks the activations travel ahead of their use: 1 (two LDS buffers of 128 rows), or 3 with four buffers of the 32 rows a
// decode-size tile has (MT = 2): there a K block is a few MFMAs, and with the activations only one block ahead every one of the
// reduction's 56 blocks (K = 7168) waited out a global -> LDS round trip (0.9 us per block, 51 us per tile, whatever the ring depth
// of the weights or the number of workgroups)
template <int MODE, int MT, bool ODD, int NW = 8, int XD = 1, bool NT = false>
SGLK_DEV void run(const MoeGemmParams& p, unsigned char* lds, const TileCtx& c) {
    static_assert(NW == 8 || MODE == MODE_GATE_UP, "narrow workgroups exist for GATE_UP only");
    // (ODD with XD = 3: a reduction of exactly three blocks -- expert width 384 in DOWN -- all of whose activations are requested up front)
    static_assert(XD == 1 || (XD == 3 && MT == 2 && ((!ODD && MODE == MODE_GATE_UP) || (ODD && MODE == MODE_DOWN))),
                  "the far prefetch is built for short tiles: GATE_UP with even block counts, DOWN with three blocks");
    constexpr int XB = XD + 1;                              // LDS buffers of the activations
    constexpr int kXSz = XD == 1 ? kXBuf : MT * 16 * 256;   // bytes per buffer
    // DOWN: the rows' output slots and routing weights are two dependent round trips; they start here and wait in LDS
    // for the epilogue instead of being fetched by it
    int my_slot = -1;
    float my_tw = 0.f;
    if (MODE == MODE_DOWN && (int)threadIdx.x < c.rows) my_slot = p.sorted_slot[c.pos0 + threadIdx.x];
    constexpr int TPW = MODE == MODE_GATE_UP ? 2 : 1;   // PLAIN (small-M dense GEMM, one "expert") is shaped like DOWN
    constexpr int PB = 2 * TPW;          // weight pieces per K block per wave
    // the weight stream starts before anything else: K blocks 0 and 1 of the wave's tiles; ring slot = block*PB + tile*2
    // + k half
    u32x4 ring[2 * PB];
#pragma unroll
    for (int i = 0; i < 2 * PB; ++i)
        ring[i] = ld_stream16<NT>(c.wp[(i % PB) >> 1] + (int64_t)(2 * (i / PB) + (i & 1)) * 1024);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    constexpr int XV = MT * 4 / NW;      // 1-KiB LDS-DMA pieces (4 rows x 256 B) of a K block per wave: MT*16 rows / 4 / NW

    // ---- activations: global -> LDS by DMA, no registers.  Piece q = wave*XV + j holds rows 4q .. 4q+3; lane l lands at
    //      byte 16*l of the piece (row 4q + (l>>4), chunk position l&15), so the swizzle is applied to the SOURCE chunk --
    const uint16_t* xsrc[XV];
#pragma unroll
    for (int j = 0; j < XV; ++j) {
        const int row = (wave * XV + j) * 4 + (lane >> 4), ch = (lane & 15) ^ (row & 15);
        // rows past the tile's last one re-read that last row: whatever they hold only reaches accumulator columns that
        // are never stored, and an unconditional load keeps control flow out of the pipeline
        const int rr = row < c.rows ? row : c.rows - 1;
        int64_t xrow;
        if (MODE == MODE_GATE_UP) xrow = (int64_t)(p.sorted_slot[c.pos0 + rr] / p.topk) * p.x_stride;
        else xrow = (int64_t)(c.pos0 + rr) * p.x_stride;
        xsrc[j] = p.x + xrow + ch * 8 + (int64_t)c.kb0 * 128;
    }
    auto x_dma = [&](int kb) __attribute__((always_inline)) {
        unsigned char* dst = lds + (kb % XB) * kXSz + wave * XV * 1024;
#pragma unroll
        for (int j = 0; j < XV; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[j] + kb * 128), (lptr_t)(dst + j * 1024), 16, 0, 0);
    };

    f32x4 acc[TPW][MT];
#pragma unroll
    for (int a = 0; a < TPW; ++a)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[a][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto piece_ptr = [&](int kb, int j) __attribute__((always_inline)) { return c.wp[j >> 1] + (int64_t)(2 * kb + (j & 1)) * 1024; };

    // one 128-wide K block: ring slots PB*half .. = (tile0,k0) (tile0,k1) [(tile1,k0) (tile1,k1)]
    // has_x: block kb+XD exists (its activations are requested here); prefetch_x: block kb+1 exists (sync at the end of this block)
    auto block = [&](int kb, int half, bool refill, bool prefetch_x, bool has_x) __attribute__((always_inline)) {
        // block kb+XD's activations first, then the weight refills: loads retire in order, so "all but the newest ones of THIS
        // block have landed" at the end of the block means the activations and the weights of block kb+1 are in
        if (has_x) x_dma(kb + XD);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 w[TPW][4];   // [tile][k-step of the block]
#pragma unroll
        for (int a = 0; a < TPW; ++a) {
            const u32x4 r0 = ring[half * PB + 2 * a], r1 = ring[half * PB + 2 * a + 1];
            w[a][0] = cvt8(r0[0], r0[1]);
            w[a][1] = cvt8(r0[2], r0[3]);
            w[a][2] = cvt8(r1[0], r1[1]);
            w[a][3] = cvt8(r1[2], r1[3]);
        }
        if (refill) {
#pragma unroll
            for (int j = 0; j < PB; ++j) ring[half * PB + j] = ld_stream16<NT>(piece_ptr(kb + 2, j));
        }
        __builtin_amdgcn_sched_barrier(0);
        float sc[TPW];
#pragma unroll
        for (int a = 0; a < TPW; ++a) sc[a] = c.sc[a * kMaxKB + kb];
        const unsigned char* xb = lds + (kb % XB) * kXSz;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int xr = mt * 16 + r;
            const unsigned char* base = xb + xr * 256;
            bf16x8 x[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) x[s] = *reinterpret_cast<const bf16x8*>(base + (((s * 4 + g) ^ (xr & 15)) << 4));
            f32x4 t[TPW];
#pragma unroll
            for (int a = 0; a < TPW; ++a) t[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int a = 0; a < TPW; ++a) t[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[a][s], x[s], t[a], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < TPW; ++a) acc[a][mt] += sc[a] * t[a];
        }
        if (prefetch_x) {   // the DMA target was last read in block kb-1, which every wave left before the barrier that closed it
            // the waits are builtins, not inline asm: the compiler's own wait-count pass sees them and knows the DMA has
            // landed; behind an opaque asm it re-waits with vmcnt(0) in front of the next LDS read and drains the ring
            if (XD == 1) {
                if (refill) __builtin_amdgcn_s_waitcnt(0x0F70 | PB);   // vmcnt(PB)
                else __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
            } else {   // this block's own requests may stay in flight: XV pieces of block kb+XD, PB weight pieces of block kb+2
                if (has_x && refill) __builtin_amdgcn_s_waitcnt(0x0F70 | (XV + PB));
                else if (refill) __builtin_amdgcn_s_waitcnt(0x0F70 | PB);
                else __builtin_amdgcn_s_waitcnt(0x0F70);
            }
            __builtin_amdgcn_s_barrier();
        }
    };

    if (MODE == MODE_DOWN && my_slot >= 0) my_tw = p.topk_weights[my_slot];
    x_dma(0);
    if (XD == 3) {   // (kblocks >= 4: launcher)
        x_dma(1);
        x_dma(2);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0): block 0's activations and the scale table
    int* slot_tab = reinterpret_cast<int*>(lds + kRowTabOff);
    float* tw_tab = reinterpret_cast<float*>(lds + kRowTabOff + kTM * 4);
    if (MODE == MODE_DOWN && tid < kTM) {
        slot_tab[tid] = my_slot;
        tw_tab[tid] = my_tw;
    }
    __syncthreads();
    // kblocks >= 2.  The steady state has no branch between a load and its use (with one the compiler falls back to
    // s_waitcnt vmcnt(0) per piece and the ring degenerates): pairs of blocks with unconditional refills, then a tail of two
    // (even count) or three (odd count, e.g. N = 384 -> 3 blocks) blocks whose flags are literals too
    int kb = 0;
    if (XD == 3 && ODD) {   // exactly three blocks (launcher), all three requested in the prologue
        block(0, 0, true, true, false);
        block(1, 1, false, true, false);
        block(2, 0, false, false, false);
    } else if (XD == 3) {   // even count >= 4: pairs with everything on, then the last four blocks with literal flags
        for (; kb + 5 <= c.kblocks; kb += 2) {
            block(kb, 0, true, true, true);
            block(kb + 1, 1, true, true, true);
        }
        block(kb, 0, true, true, true);
        block(kb + 1, 1, true, true, false);
        block(kb + 2, 0, false, true, false);
        block(kb + 3, 1, false, false, false);
    } else if (!ODD) {   // the parity of the block count is a template parameter: both tails in one kernel cost registers
        for (; kb + 2 < c.kblocks; kb += 2) {
            block(kb, 0, true, true, true);
            block(kb + 1, 1, true, true, true);
        }
        block(kb, 0, false, true, true);
        block(kb + 1, 1, false, false, false);
    } else {
        for (; kb + 3 < c.kblocks; kb += 2) {
            block(kb, 0, true, true, true);
            block(kb + 1, 1, true, true, true);
        }
        block(kb, 0, true, true, true);
        block(kb + 1, 1, false, true, true);
        block(kb + 2, 0, false, false, false);
    }

    // ---- epilogue: lane holds weight rows 4g..4g+3 of each tile for token r of every column tile ----------------------
    const int q4 = g * 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int tr = mt * 16 + r;
        if (tr >= c.rows) continue;
        if (MODE == MODE_GATE_UP) {
            const f32x4 gt = acc[0][mt], up = acc[TPW - 1][mt];
            uint2 v;
            v.x = pack_bf16x2(silu_f32(gt[0]) * up[0], silu_f32(gt[1]) * up[1]);
            v.y = pack_bf16x2(silu_f32(gt[2]) * up[2], silu_f32(gt[3]) * up[3]);
            *reinterpret_cast<uint2*>(p.out + (int64_t)(c.pos0 + tr) * p.out_stride + c.ntile * (NW * 16) + wave * 16 + q4) = v;
        } else if (MODE == MODE_PLAIN) {
            const int col = c.ntile * 128 + wave * 16 + q4;
            const f32x4 v4 = acc[0][mt];
            if (p.partial) {      // fp32 partial of this K range; bias and the bf16 rounding belong to the ordered reduce
                float* dst = p.partial + ((int64_t)c.ksr * p.split_rows + c.pos0 + tr) * p.out_cols + col;
                *reinterpret_cast<float4*>(dst) = make_float4(v4[0], v4[1], v4[2], v4[3]);
            } else {
                float b4[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + col); b4[0] = b.x; b4[1] = b.y; b4[2] = b.z; b4[3] = b.w; }
                uint2 v;
                v.x = pack_bf16x2(v4[0] + b4[0], v4[1] + b4[1]);
                v.y = pack_bf16x2(v4[2] + b4[2], v4[3] + b4[3]);
                *reinterpret_cast<uint2*>(p.out + (int64_t)(c.pos0 + tr) * p.out_stride + col) = v;
            }
        } else {
            const int slot = slot_tab[tr];
            const float tw = tw_tab[tr];
            uint16_t* orow = p.out + (int64_t)slot * p.out_stride + c.ntile * 128 + wave * 16 + q4;
            const f32x4 v4 = acc[0][mt] * tw;
            uint2 v;
            v.x = pack_bf16x2(v4[0], v4[1]);
            v.y = pack_bf16x2(v4[2], v4[3]);
            *reinterpret_cast<uint2*>(orow) = v;
        }
    }
}

// DOWN over TWO neighbouring column tiles (2 x 128 output columns) of one m-tile per workgroup, even block counts: the weight
// ring runs straight from the first tile's last K blocks into the second tile's first ones, so the pair pays ONE prologue and
// one ring fill (a DOWN tile at N = 768 is only six blocks long); the activations are simply streamed through LDS twice.
template <int MT, bool NT>
SGLK_DEV void run_down2(const MoeGemmParams& p, unsigned char* lds, const TileCtx& c) {
    int my_slot = -1;
    float my_tw = 0.f;
    if ((int)threadIdx.x < c.rows) my_slot = p.sorted_slot[c.pos0 + threadIdx.x];
    const int nk = c.kblocks;
    // virtual block vb = tile * nk + kb; c.wp[0] / c.wp[1] = the wave's row tile in the first / second column tile
    auto wptr = [&](int vb) __attribute__((always_inline)) {
        const bool second = vb >= nk;
        return (second ? c.wp[1] : c.wp[0]) + (int64_t)(2 * (second ? vb - nk : vb)) * 1024;
    };
    u32x4 ring[4];   // slot = (block & 1) * 2 + k half
#pragma unroll
    for (int i = 0; i < 4; ++i) ring[i] = ld_stream16<NT>(wptr(i >> 1) + (i & 1) * 1024);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    constexpr int XV = MT / 2;
    const uint16_t* xsrc[XV];
#pragma unroll
    for (int j = 0; j < XV; ++j) {
        const int row = (wave * XV + j) * 4 + (lane >> 4), ch = (lane & 15) ^ (row & 15);
        const int rr = row < c.rows ? row : c.rows - 1;
        xsrc[j] = p.x + (int64_t)(c.pos0 + rr) * p.x_stride + ch * 8;
    }
    auto x_dma = [&](int buf, int kb) __attribute__((always_inline)) {
        unsigned char* dst = lds + buf * kXBuf + wave * XV * 1024;
#pragma unroll
        for (int j = 0; j < XV; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[j] + kb * 128), (lptr_t)(dst + j * 1024), 16, 0, 0);
    };
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // block kb of column tile `tile` (virtual block vb = tile * nk + kb; nk is even, so vb & 1 == kb & 1 == half)
    auto block = [&](int tile, int kb, int half, bool refill, bool prefetch_x) __attribute__((always_inline)) {
        const int vb = tile * nk + kb;
        if (prefetch_x) x_dma(half ^ 1, kb + 1 == nk ? 0 : kb + 1);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 w[4];
        {
            const u32x4 r0 = ring[half * 2], r1 = ring[half * 2 + 1];
            w[0] = cvt8(r0[0], r0[1]);
            w[1] = cvt8(r0[2], r0[3]);
            w[2] = cvt8(r1[0], r1[1]);
            w[3] = cvt8(r1[2], r1[3]);
        }
        if (refill) {
            const unsigned char* nx = wptr(vb + 2);
            ring[half * 2] = ld_stream16<NT>(nx);
            ring[half * 2 + 1] = ld_stream16<NT>(nx + 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float sc = c.sc[tile * kMaxKB + kb];
        const unsigned char* xb = lds + half * kXBuf;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int xr = mt * 16 + r;
            const unsigned char* base = xb + xr * 256;
            bf16x8 x[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) x[s] = *reinterpret_cast<const bf16x8*>(base + (((s * 4 + g) ^ (xr & 15)) << 4));
            f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[s], x[s], t, 0, 0, 0);
            acc[mt] += sc * t;
        }
        if (prefetch_x) {
            if (refill) __builtin_amdgcn_s_waitcnt(0x0F70 | 2);   // vmcnt(2): all but the two refills -> the DMA has landed
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();
        }
    };
    auto store = [&](int tile) __attribute__((always_inline)) {
        const int* slot_tab = reinterpret_cast<const int*>(lds + kRowTabOff);
        const float* tw_tab = reinterpret_cast<const float*>(lds + kRowTabOff + kTM * 4);
        const int col = (c.ntile * 2 + tile) * 128 + wave * 16 + g * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int tr = mt * 16 + r;
            if (tr < c.rows) {
                const f32x4 v4 = acc[mt] * tw_tab[tr];
                uint2 v;
                v.x = pack_bf16x2(v4[0], v4[1]);
                v.y = pack_bf16x2(v4[2], v4[3]);
                *reinterpret_cast<uint2*>(p.out + (int64_t)slot_tab[tr] * p.out_stride + col) = v;
            }
            acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };

    if (my_slot >= 0) my_tw = p.topk_weights[my_slot];
    x_dma(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0070);
    if (tid < kTM) {
        reinterpret_cast<int*>(lds + kRowTabOff)[tid] = my_slot;
        reinterpret_cast<float*>(lds + kRowTabOff + kTM * 4)[tid] = my_tw;
    }
    __syncthreads();
    // first column tile: every block refills (the stream continues into the second tile) and prefetches (block nk-1 fetches
    // the activations' block 0 again)
    for (int kb = 0; kb < nk; kb += 2) {
        block(0, kb, 0, true, true);
        block(0, kb + 1, 1, true, true);
    }
    store(0);
    int kb = 0;
    for (; kb + 2 < nk; kb += 2) {
        block(1, kb, 0, true, true);
        block(1, kb + 1, 1, true, true);
    }
    block(1, kb, 0, false, true);
    block(1, kb + 1, 1, false, false);
    store(1);
}

template <int MODE, bool ODD, int NW = 8, int XD = 1, bool NT = false>
__global__ __launch_bounds__(NW * 64, MODE == MODE_GATE_UP ? 2 : 4) void moe_gemm_fp8w_mid_kernel(const MoeGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nsplit = (MODE == MODE_PLAIN && p.ksplit > 1) ? p.ksplit : 1;
    // dense (PLAIN without a tile table): the m-tiles are rows [96 i, 96 i + 96) of split_rows rows -- nothing to look up,
    // which takes two dependent loads off the front of a workgroup that only lives for a few K blocks
    const bool dense = MODE == MODE_PLAIN && p.tile_info == nullptr;
    const int mtiles = dense ? (p.split_rows + kTM - 1) / kTM : p.num_tiles[0];
    const int live = mtiles * p.n_tiles * nsplit;
    if ((int)blockIdx.x >= live) return;
    const int Ls = xcd_remap(blockIdx.x, live);
    const int L = Ls / nsplit;
    const int mtile = L / p.n_tiles;
    int4 ti;
    if (dense) {
        const int r0 = mtile * kTM;
        ti = make_int4(0, r0, p.split_rows - r0 < kTM ? p.split_rows - r0 : kTM, 0);
    } else {
        ti = p.tile_info[mtile];
    }
    const int e = __builtin_amdgcn_readfirstlane(ti.x);

    TileCtx c;
    c.ntile = L - mtile * p.n_tiles;
    c.pos0 = __builtin_amdgcn_readfirstlane(ti.y);
    c.rows = __builtin_amdgcn_readfirstlane(ti.z);
    c.ksr = Ls - L * nsplit;
    c.kblocks = nsplit > 1 ? p.split_kblocks : p.C >> 7;
    c.kb0 = c.ksr * c.kblocks;
    const int ctiles = p.C >> 6;
    if (MODE == MODE_GATE_UP) {          // workgroup = NW * 16 ic1 columns: wave w -> columns ntile * (NW * 16) + 16w .. +15
        c.row16[0] = c.ntile * NW + wave;
        c.row16[1] = (p.n_half >> 4) + c.ntile * NW + wave;
    } else {                             // DOWN / PLAIN: 128 output columns, wave w -> columns ntile*128 + 16w .. +15
        c.row16[0] = c.ntile * 8 + wave;
        c.row16[1] = c.row16[0];
    }
    const unsigned char* wexp = p.w + (int64_t)e * p.w_expert_stride;
    c.wp[0] = wexp + ((int64_t)c.row16[0] * ctiles + 2 * c.kb0) * 1024 + lane * 16;
    c.wp[1] = wexp + ((int64_t)c.row16[1] * ctiles + 2 * c.kb0) * 1024 + lane * 16;

    // block scales of the wave's two tiles -> LDS (read back as broadcasts, one per K block)
    float* sc = reinterpret_cast<float*>(lds + kScOff) + wave * (2 * kMaxKB);
    {
        const float* scale_e = p.w_scale + (int64_t)e * p.scale_rows * p.scale_cols;
#pragma unroll
        for (int a = 0; a < 2; ++a) {       // lane = K block of the workgroup's range
            const int srow = ((a ? c.row16[1] : c.row16[0]) * 16) / p.block_n;
            sc[a * kMaxKB + lane] = lane < c.kblocks ? scale_e[srow * p.scale_cols + c.kb0 + lane] : 0.f;
        }
    }
    c.sc = sc;

    const int mt = (c.rows + 15) >> 4;
    if (mt <= 2) run<MODE, 2, ODD, NW, XD, NT>(p, lds, c);
    else if (mt <= 4) run<MODE, 4, ODD, NW, 1, NT>(p, lds, c);
    else run<MODE, 6, ODD, NW, 1, NT>(p, lds, c);
}

// DOWN, two column tiles per workgroup (p.n_tiles = output columns / 256)
template <bool NT>
__global__ __launch_bounds__(512, 4) void moe_gemm_fp8w_mid_down2_kernel(const MoeGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int live = p.num_tiles[0] * p.n_tiles;
    if ((int)blockIdx.x >= live) return;
    const int L = xcd_remap(blockIdx.x, live);
    const int mtile = L / p.n_tiles;
    const int4 ti = p.tile_info[mtile];
    const int e = __builtin_amdgcn_readfirstlane(ti.x);
    TileCtx c;
    c.ntile = L - mtile * p.n_tiles;
    c.pos0 = __builtin_amdgcn_readfirstlane(ti.y);
    c.rows = __builtin_amdgcn_readfirstlane(ti.z);
    c.kblocks = p.C >> 7;
    c.kb0 = 0;
    c.ksr = 0;
    const int ctiles = p.C >> 6;
    c.row16[0] = c.ntile * 16 + wave;        // first column tile: columns ntile*256 + 16 wave ..
    c.row16[1] = c.row16[0] + 8;             // second: 128 columns further
    const unsigned char* wexp = p.w + (int64_t)e * p.w_expert_stride;
    c.wp[0] = wexp + ((int64_t)c.row16[0] * ctiles) * 1024 + lane * 16;
    c.wp[1] = wexp + ((int64_t)c.row16[1] * ctiles) * 1024 + lane * 16;
    float* sc = reinterpret_cast<float*>(lds + kScOff) + wave * (2 * kMaxKB);
    {
        const float* scale_e = p.w_scale + (int64_t)e * p.scale_rows * p.scale_cols;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int srow = (c.row16[a] * 16) / p.block_n;
            sc[a * kMaxKB + lane] = lane < c.kblocks ? scale_e[srow * p.scale_cols + lane] : 0.f;
        }
    }
    c.sc = sc;
    const int mt = (c.rows + 15) >> 4;
    if (mt <= 2) run_down2<2, NT>(p, lds, c);
    else if (mt <= 4) run_down2<4, NT>(p, lds, c);
    else run_down2<6, NT>(p, lds, c);
}

}  // namespace gmid

// DOWN with two column tiles per workgroup: even K-block counts and an even number of 128-column tiles
int launch_moe_gemm_fp8w_mid_down2(const MoeGemmParams& p, int max_mtiles, hipStream_t stream) {
    const int kblocks = p.C >> 7;
    if (p.C % 256 != 0 || kblocks < 2 || kblocks > gmid::kMaxKB) SGLK_FAIL(SGLK_ERR_SHAPE, "moe_gemm_fp8w_mid_down2: reduction length %d", p.C);
    const int64_t blocks = (int64_t)max_mtiles * p.n_tiles;
    if (blocks == 0) return SGLK_OK;
    if (p.w_nt) {
        SGLK_ENSURE_DYN_LDS(gmid::moe_gemm_fp8w_mid_down2_kernel<true>, gmid::kLds, "moe_gemm_fp8w_mid_down2");
        hipLaunchKernelGGL(gmid::moe_gemm_fp8w_mid_down2_kernel<true>, dim3((unsigned)blocks), dim3(512), gmid::kLds, stream, p);
    } else {
        SGLK_ENSURE_DYN_LDS(gmid::moe_gemm_fp8w_mid_down2_kernel<false>, gmid::kLds, "moe_gemm_fp8w_mid_down2");
        hipLaunchKernelGGL(gmid::moe_gemm_fp8w_mid_down2_kernel<false>, dim3((unsigned)blocks), dim3(512), gmid::kLds, stream, p);
    }
    SGLK_CHECK_LAUNCH("moe_gemm_fp8w_mid_down2");
    return SGLK_OK;
}

int launch_moe_gemm_fp8w_mid(int mode, const MoeGemmParams& p, int max_mtiles, hipStream_t stream) {
    const int nsplit = (mode == MODE_PLAIN && p.ksplit > 1) ? p.ksplit : 1;
    const int64_t blocks = (int64_t)max_mtiles * p.n_tiles * nsplit;
    if (blocks == 0) return SGLK_OK;
    const int kblocks = nsplit > 1 ? p.split_kblocks : p.C >> 7;
    if (p.C % 128 != 0 || kblocks < 2 || kblocks > gmid::kMaxKB || (nsplit > 1 && (p.C >> 7) != nsplit * kblocks))
        SGLK_FAIL(SGLK_ERR_SHAPE, "moe_gemm_fp8w_mid: reduction length %d / K range of %d blocks not supported", p.C, kblocks);
    if ((nsplit > 1 && !p.partial) || (p.partial && (p.out_cols <= 0 || p.split_rows <= 0)))
        SGLK_FAIL(SGLK_ERR_INVALID, "moe_gemm_fp8w_mid: split-K without a partial buffer");
    const size_t lds = gmid::kLds;
#define MID_LAUNCH3(MD, OD, XDV, NTV)                                                                              \
    {                                                                                                              \
        SGLK_ENSURE_DYN_LDS((gmid::moe_gemm_fp8w_mid_kernel<MD, OD, 8, XDV, NTV>), lds, "moe_gemm_fp8w_mid");      \
        hipLaunchKernelGGL((gmid::moe_gemm_fp8w_mid_kernel<MD, OD, 8, XDV, NTV>), dim3((unsigned)blocks), dim3(512), lds, stream, p); \
    }
// dense (PLAIN) launches keep the default policy: non-temporal reads made fp8_scaled_mm at 1 ... 96 rows 0-10 % slower on weights of
// 8-25 MB, replayed or rotating (tools/dense_probe.py, profiles/r03_ab_nt_weights.txt)
#define MID_LAUNCH2(MD, OD)                                                                                        \
    {                                                                                                              \
        if (MD != MODE_PLAIN && p.w_nt) MID_LAUNCH3(MD, OD, 1, (MD != MODE_PLAIN)) else MID_LAUNCH3(MD, OD, 1, false)  \
    }
#define MID_LAUNCH(MD)                                                                                             \
    {                                                                                                              \
        if (kblocks & 1) MID_LAUNCH2(MD, true) else MID_LAUNCH2(MD, false)                                         \
    }
    // GATE_UP launches that reach only part of the chip (decode-size batches; same-box A/B at the reference's decode shape, M = 1 ... 32,
    // profiles/r03_ab_mid_decode.txt): up to half the CUs -> 64 ic1 columns per workgroup (four waves) instead of 128; up to two
    // rounds of the chip and an even block count -> the activations three K blocks ahead (run<>: NW, XD).  GEMM-1 at M = 1 / 4 / 8 /
    // 16: 39.7 -> 27.7 / 51.4 -> 35.8 / 62.2 -> 59.6 / 125 -> 121 us.  SGLK_MID_NW / SGLK_MID_FAR force, SGLK_NO_MID_NARROW = as before.
    int nw = 8;
    const int cus = device_cu_count();
    if (mode == MODE_GATE_UP && !knobs().no_mid_narrow) {
        if (blocks * 2 <= cus) nw = 4;
        if (knobs().mid_nw == 4 || knobs().mid_nw == 8) nw = knobs().mid_nw;
    }
    const bool far = mode == MODE_GATE_UP && !knobs().no_mid_narrow && !(kblocks & 1) && kblocks >= 4 &&
                     (knobs().mid_far >= 0 ? knobs().mid_far == 1 : blocks < 2 * (int64_t)cus);
    if (nw != 8 || far) {
        MoeGemmParams q = p;
        q.n_tiles = p.n_tiles * (8 / nw);
        const int64_t nb = (int64_t)max_mtiles * q.n_tiles;
#define MID_NARROW(OD, NWV, XDV)                                                                                                \
    {                                                                                                                           \
        SGLK_ENSURE_DYN_LDS((gmid::moe_gemm_fp8w_mid_kernel<MODE_GATE_UP, OD, NWV, XDV>), lds, "moe_gemm_fp8w_mid");            \
        hipLaunchKernelGGL((gmid::moe_gemm_fp8w_mid_kernel<MODE_GATE_UP, OD, NWV, XDV>), dim3((unsigned)nb), dim3(NWV * 64), lds, stream, q); \
    }
        if (nw == 8) MID_NARROW(false, 8, 3)
        else if (kblocks & 1) MID_NARROW(true, 4, 1)
        else if (far) MID_NARROW(false, 4, 3)
        else MID_NARROW(false, 4, 1)
#undef MID_NARROW
    } else if (mode == MODE_GATE_UP) MID_LAUNCH(MODE_GATE_UP)
    else if (mode == MODE_DOWN && kblocks == 3 && !knobs().no_mid_narrow) {   // short tiles request their three K blocks of activations up front
        if (p.w_nt) MID_LAUNCH3(MODE_DOWN, true, 3, true) else MID_LAUNCH3(MODE_DOWN, true, 3, false)
    } else if (mode == MODE_DOWN) MID_LAUNCH(MODE_DOWN)
    else MID_LAUNCH(MODE_PLAIN)
#undef MID_LAUNCH
#undef MID_LAUNCH2
#undef MID_LAUNCH3
    SGLK_CHECK_LAUNCH("moe_gemm_fp8w_mid");
    return SGLK_OK;
}

// Small-M fp8 dense GEMMs (decode-size linear layers) on the mid kernel: M <= 192 rows in tiles of up to 96, 128 output
// columns per workgroup, and the reduction cut into equal ranges of an even number of 128-wide blocks until about 512
// workgroups exist (two per CU).  Returns 0 when the shape is not taken.
int mid_dense_ksplit(int M, int N, int K) {
    if (M <= 0 || M >= kMidDenseMaxM || N % 128 != 0 || K % 256 != 0) return 0;   // capability; the dispatch policy is dense_prefers_mid
    const int kblocks = K >> 7;
    const int64_t tiles = (int64_t)ceil_div(M, kMidTileM) * (N / 128);
    if (knobs().mid_dense_model > 0) return splitk_by_rounds(kblocks, tiles, M, N, device_cu_count() * (knobs().mid_dense_model == 1 ? 2 : 1), 32);
    int best = 0;
    const int per_min = kblocks >= 4 ? 4 : 2;               // a range shorter than 4 blocks is all prologue
    for (int per = kblocks; per >= per_min; per -= 2) {    // longest ranges first
        if (kblocks % per != 0 || per > 32) continue;
        const int ks = kblocks / per;
        if (ks > 32 || (int64_t)ks * M * N * 4 > (64ll << 20)) continue;
        best = ks;
        if (tiles * ks >= 512) break;
    }
    return best;
}

}  // namespace sglk
