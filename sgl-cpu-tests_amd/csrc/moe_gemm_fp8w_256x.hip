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

#include "sglk_common.h"
#include "moe_internal.h"

namespace sglk {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

namespace g256x {

constexpr int kBM = 256;
constexpr int kStageX = kBM * 128;        // 32 KiB
constexpr int kStageW = 16 * 1024;        // 16 KiB: 16 packed 16x64 fp8 tiles
constexpr int kStage = kStageX + kStageW; // 48 KiB
constexpr int kRing = 3;
constexpr int kScaleOff = kRing * kStage; // 144 KiB, then the scale table
constexpr int kMaxKBlocks = 64;           // reduction length <= 8192
constexpr int kRowTabOff = kScaleOff + 16 * kMaxKBlocks * 4;   // + 4 KiB, then the per-row tables of the DOWN epilogue
constexpr int kLds = kRowTabOff + 2 * kBM * 4;           // + 2 KiB: output slot and routing weight per tile row

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

template <int MODE, int RESCALE>
__global__ __launch_bounds__(512, 2) void moe_gemm_fp8w_256x_kernel(const MoeGemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[kLds];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wm = wave >> 2;
#ifdef SGLK_DEV_ABLATE
    const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
#endif

    // the grid is sized for the worst case; only the first num_tiles*n_tiles blocks have work.  The XCD remap is
    // taken over THAT count, so every XCD gets an equal contiguous share of the real tiles.
    const int live = p.num_tiles[0] * p.n_tiles;
    if ((int)blockIdx.x >= live) return;
    const int L = xcd_remap(blockIdx.x, live);
    const int mtile = L / p.n_tiles;
    const int ntile = L - mtile * p.n_tiles;
    const int4 ti = p.tile_info[mtile];
    const int e = (RESCALE & 32) ? 0 : __builtin_amdgcn_readfirstlane(ti.x);   // bit 5: timing ablation, all-L2-hit operands
    const int pos0 = (RESCALE & 32) ? 0 : __builtin_amdgcn_readfirstlane(ti.y);
    const int rows = __builtin_amdgcn_readfirstlane(ti.z);

    const int ctiles = p.C >> 6;
    const int kblocks = p.C >> 7;
    const int T = ctiles;

    // workgroup's 16 weight row-tiles: GATE_UP = 8 gate + 8 up, DOWN = 16 consecutive
    auto piece_row16 = [&](int piece) {
        if (MODE == MODE_GATE_UP) return (piece < 8) ? ntile * 8 + piece : (p.n_half >> 4) + ntile * 8 + (piece - 8);
        return ntile * 16 + piece;
    };

    // ---- prologue loads.  Everything the tile needs besides the operands is FETCHED here but parked in registers;
    //      the LDS copies are written after the first LDS-DMA stages have been issued, so no operand load waits
    //      behind these round trips ---------------------------------------------------------------------------------
    float* sc = reinterpret_cast<float*>(smem + kScaleOff);            // sc[piece][kb]
    int* slot_tab = reinterpret_cast<int*>(smem + kRowTabOff);         // DOWN epilogue: output row (slot) ...
    float* tw_tab = reinterpret_cast<float*>(smem + kRowTabOff + kBM * 4);   // ... and routing weight per tile row
    float sc_reg[2] = {0.f, 0.f};                                      // 16 * kblocks <= 1024 table entries
    {
        const float* scale_e = p.w_scale + (int64_t)e * p.scale_rows * p.scale_cols;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * 512;
            if (i < 16 * kblocks) {
                const int piece = i / kblocks, kb = i - piece * kblocks;
                sc_reg[j] = scale_e[((piece_row16(piece) * 16) / p.block_n) * p.scale_cols + kb];
            }
        }
    }
    int my_slot = -1;
    if (MODE == MODE_DOWN && tid < rows) my_slot = p.sorted_slot[pos0 + tid];

    // ---- LDS-DMA sources: buffer descriptors (SGPRs) + ONE 32-bit per-lane offset per piece, fixed for the whole
    //      tile; the stage offset goes in the scalar soffset, so a stage costs 6 buffer_load...lds and no VALU ----
    const unsigned xbytes = (unsigned)__builtin_amdgcn_readfirstlane((int)p.x_bytes);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, xbytes, 0x00020000);
    const unsigned char* wexp = p.w + (int64_t)e * p.w_expert_stride;
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)wexp, 0, (unsigned)p.w_expert_stride, 0x00020000);
    unsigned xsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int rr = r < rows ? r : 0;
        int64_t xrow;
        if (MODE == MODE_GATE_UP) {
            const int slot = p.sorted_slot[pos0 + rr];
            xrow = (int64_t)(slot / p.topk) * p.x_stride;
        } else {
            xrow = (int64_t)(pos0 + rr) * p.x_stride;
        }
        xsrc[i] = (unsigned)(xrow * 2) + (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) << 4);   // image swizzle: chunk ^ ((row>>1)&7)
    }
    unsigned wsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) wsrc[i] = (unsigned)(piece_row16(wave * 2 + i) * ctiles) * 1024u + lane * 16;

    // one of the wave's six 1-KiB LDS-DMA pieces of stage kt (0..3: X rows, 4..5: packed W tiles).  The pieces of a
    // stage are issued ONE PER MFMA GROUP, never as a burst: a buffer_load...lds costs the issuing wave 60-180 cycles,
    // which hides behind the MFMAs already queued on the matrix pipe but stalls the wave when six come back to back.
    auto issue_piece = [&](int kt, int buf, int i) {
        if ((RESCALE & 64) && kt > 2) return;   // bit 6: timing ablation, no LDS-DMA in the steady state
        unsigned char* sx = smem + buf * kStage;
        if (i < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lptr_t)(sx + (wave * 4 + i) * 1024), 16, xsrc[i], kt * 128, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lptr_t)(sx + kStageX + (wave * 2 + i - 4) * 1024), 16,
                                                     wsrc[i - 4], kt * 1024, 0, 0);
    };
    auto issue_stage = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < 6; ++i) issue_piece(kt, buf, i);
    };

    // ---- operand addressing for mfma_f32_32x32x16_bf16 (A = weights, B = tokens) ---------------------------------------
    // lane l: h = l>>5 (k octet inside the 16-wide k-step), r = l&31 (operand row / column).
    // Weight row tile rt (32 rows) = two packed 16-row pieces; octet o = 2*ks + h of row r sits in piece (r>>4) at lane
    // slot ((o&3)*16 + (r&15)), byte (o>>2)*8 (pack.hip order) -> one ds_read_b64 per (rt, ks).
    const int h = lane >> 5, r32 = lane & 31;
    int wbase[2];      // byte offset of (rt, lane) inside a W stage, without the k-step term
    int wpiece0[2];    // first piece of the row tile (for the scale table)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        if (MODE == MODE_GATE_UP) wpiece0[rt] = (rt == 0) ? wn * 2 : 8 + wn * 2;
        else wpiece0[rt] = wn * 4 + rt * 2;
        wbase[rt] = (wpiece0[rt] + (r32 >> 4)) * 1024 + (r32 & 15) * 16;
    }
    // k-step ks: octet o = 2ks + h -> slot group (o&3), half (o>>2)
    auto woff = [&](int rt, int ks) { return wbase[rt] + (((2 * ks + h) & 3) * 16) * 16 + ((2 * ks + h) >> 2) * 8; };
    // token tile tt (32 tokens): row = wm*128 + tt*32 + r32, chunk 2ks + h, swizzled by (row>>1)&7
    const int xrow0 = wm * 128 + r32;
    auto xoff = [&](int tt, int ks) {
        const int row = xrow0 + tt * 32;
        return row * 128 + (((2 * ks + h) ^ ((row >> 1) & 7)) << 4);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][tt][i] = 0.f;

    float pow2[2], pow2_next[2], mant[2], ratio[2];

    // prologue: three stages in flight, wait for the first
    issue_stage(0, 0);
    if (T > 1) issue_stage(1, 1);
    if (T > 2) issue_piece(2, 2, 0);
    {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * 512;
            if (i < 16 * kblocks) {
                const int piece = i / kblocks, kb = i - piece * kblocks;
                sc[piece * kMaxKBlocks + kb] = sc_reg[j];
            }
        }
        if (MODE == MODE_DOWN && tid < kBM) {
            slot_tab[tid] = my_slot;
            tw_tab[tid] = my_slot >= 0 ? p.topk_weights[my_slot] : 0.f;
        }
    }
    __syncthreads();   // tables visible; drains the prologue DMA once

#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        split_scale(sc[wpiece0[rt] * kMaxKBlocks], pow2[rt], mant[rt]);
        pow2_next[rt] = pow2[rt];
        ratio[rt] = 1.f;
    }

    // A stage = 4 k-steps x 2 token-tile pairs = 8 groups of 4 MFMAs (2 row tiles x 2 token tiles).  The X fragments
    // of the next group and the raw weight octets of the next k-step are read from LDS while the current group's
    // MFMAs run.
    bf16x8 xa[2], xb[2], wf[2];
    u32x2 wraw[2];
    auto read_x = [&](bf16x8 (&xf)[2], int buf, int ks, int tp) {
        if ((RESCALE & 16) && (ks | tp)) return;   // timing ablation only: one X read per stage
        const unsigned char* sx = smem + buf * kStage;
        xf[0] = *reinterpret_cast<const bf16x8*>(sx + xoff(2 * tp, ks));
        xf[1] = *reinterpret_cast<const bf16x8*>(sx + xoff(2 * tp + 1, ks));
    };
    auto read_w = [&](int buf, int ks) {
        const unsigned char* sw = smem + buf * kStage + kStageX;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) wraw[rt] = *reinterpret_cast<const u32x2*>(sw + woff(rt, ks));
    };
    auto cvt_w = [&]() {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            if (RESCALE & 8) {   // timing ablation only: no conversion
                u32x4 t = {wraw[rt][0], wraw[rt][1], wraw[rt][0], wraw[rt][1]};
                wf[rt] = __builtin_bit_cast(bf16x8, t);
            } else {
                wf[rt] = cvt8(wraw[rt][0], wraw[rt][1], pow2[rt]);
            }
        }
    };
    auto group = [&](const bf16x8 (&xf)[2], int tp) {
        if (RESCALE & 2048) __builtin_amdgcn_s_setprio(1);   // per-group priority flips measured 3-4 % SLOWER here: off
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[rt][2 * tp + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[rt], xf[j], acc[rt][2 * tp + j], 0, 0, 0);
        if (RESCALE & 2048) __builtin_amdgcn_s_setprio(0);
    };

    read_w(0, 0);
    read_x(xa, 0, 0, 0);
    if ((RESCALE & 512) && __builtin_amdgcn_readfirstlane(tid) >= 256) __builtin_amdgcn_s_setprio(1);   // ablation: static priority for the younger half

    int buf = 0;
    const bool active = wm * 128 < rows;
    auto idle_stage = [&](int t, bool wait6, bool dma2, int dma3, bool dma3_ok) {
        int pbuf = buf - 1;
        if (pbuf < 0) pbuf = kRing - 1;
        if (dma2) {
#pragma unroll
            for (int i = 1; i < 6; ++i) issue_piece(t + 2, pbuf, i);
        }
        if (wait6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (dma3 == 1 || (dma3 == 2 && dma3_ok)) issue_piece(t + 3, buf, 0);
        buf = (buf + 1 == kRing) ? 0 : buf + 1;
    };
#ifdef SGLK_DEV_ABLATE
    unsigned long long dma_wait = 0, bar_wait = 0;
#endif
    // One 64-deep stage t; every flag is a literal at the call site (no control flow around the MFMA groups).
    auto stage = [&](int t, bool first, bool closing, bool wait6, bool dma2, int dma3, bool dma3_ok, bool more) {
        int nbuf = buf + 1;
        if (nbuf == kRing) nbuf = 0;
        int pbuf = buf - 1;
        if (pbuf < 0) pbuf = kRing - 1;
        if (first) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) pow2[rt] = pow2_next[rt];
            // K block boundary: the whole accumulator into units of the new mantissa, in place
            if (!(RESCALE & 4)) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) acc[rt][tt] *= ratio[rt];
            }
        }
        if (closing) {
            const int kb = (t + 1) >> 1;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                float nm;
                split_scale(sc[wpiece0[rt] * kMaxKBlocks + kb], pow2_next[rt], nm);
                ratio[rt] = uniform_f32(mant[rt] * __builtin_amdgcn_rcpf(nm));
                mant[rt] = nm;
            }
        }
        // ks = 0
        read_x(xb, buf, 0, 1);
        cvt_w();
        read_w(buf, 1);
        group(xa, 0);
        if (dma2) issue_piece(t + 2, pbuf, 1);
        read_x(xa, buf, 1, 0);
        group(xb, 1);
        if (dma2) issue_piece(t + 2, pbuf, 2);
        // ks = 1
        read_x(xb, buf, 1, 1);
        cvt_w();
        read_w(buf, 2);
        group(xa, 0);
        if (dma2) issue_piece(t + 2, pbuf, 3);
        read_x(xa, buf, 2, 0);
        group(xb, 1);
        if (dma2) issue_piece(t + 2, pbuf, 4);
        // ks = 2
        read_x(xb, buf, 2, 1);
        cvt_w();
        read_w(buf, 3);
        group(xa, 0);
        if (dma2) issue_piece(t + 2, pbuf, 5);
        read_x(xa, buf, 3, 0);
        group(xb, 1);
        // ks = 3
        read_x(xb, buf, 3, 1);
        cvt_w();
        group(xa, 0);
        // sync point S_t in front of the last group (see the 16x16 kernel): stage t's buffer is free for the DMA of
        // stage t+3, stage t+1 has landed, its first fragments are read now
#ifdef SGLK_DEV_ABLATE
        unsigned long long ta = 0, tb = 0;
        if (p.dbg) ta = __builtin_amdgcn_s_memtime();
#endif
        if (wait6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SGLK_DEV_ABLATE
        if (p.dbg) tb = __builtin_amdgcn_s_memtime();
#endif
        if (RESCALE & 1024) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ablation: NO barrier (racy, timing only)
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef SGLK_DEV_ABLATE
        if (p.dbg) { const unsigned long long tc = __builtin_amdgcn_s_memtime(); dma_wait += tb - ta; bar_wait += tc - tb; }
#endif
        if (dma3 == 1 || (dma3 == 2 && dma3_ok)) issue_piece(t + 3, buf, 0);
        if (more) {
            read_w(nbuf, 0);
            read_x(xa, nbuf, 0, 0);
        }
        group(xb, 1);
        buf = nbuf;
    };
#ifdef SGLK_DEV_ABLATE
    // in-kernel clock of the main loop: shader cycles (s_memtime) over constant 100 MHz ticks (s_memrealtime)
    unsigned long long t0 = 0, r0 = 0;
    if (p.dbg && tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    if (active) {
        for (int kb = 0; kb + 1 < kblocks; ++kb) {
            stage(2 * kb, true, false, true, true, 1, true, true);            // ratio == 1 for kb == 0
            stage(2 * kb + 1, false, true, true, true, 2, kb + 2 < kblocks, true);
        }
        stage(T - 2, true, false, false, false, 0, false, true);
        stage(T - 1, false, false, false, false, 0, false, false);
    } else {
        for (int kb = 0; kb + 1 < kblocks; ++kb) {
            idle_stage(2 * kb, true, true, 1, true);
            idle_stage(2 * kb + 1, true, true, 2, kb + 2 < kblocks);
        }
        idle_stage(T - 2, false, false, 0, false);
        idle_stage(T - 1, false, false, 0, false);
    }
#ifdef SGLK_DEV_ABLATE
    if (p.dbg && tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        p.dbg[32 * blockIdx.x] = t1 - t0;
        p.dbg[32 * blockIdx.x + 1] = r1 - r0;
        p.dbg[32 * blockIdx.x + 18] = rt_entry;   // absolute 100 MHz ticks: entry, loop start, loop end
        p.dbg[32 * blockIdx.x + 19] = r0;
        p.dbg[32 * blockIdx.x + 20] = r1;
        p.dbg[32 * blockIdx.x + 22] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
        p.dbg[32 * blockIdx.x + 23] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
    }
    if (p.dbg && lane == 0) {
        p.dbg[32 * blockIdx.x + 2 + wave] = dma_wait;
        p.dbg[32 * blockIdx.x + 10 + wave] = bar_wait;
    }
#endif

    // ---- epilogue: accumulator -> LDS image [token][column] (16-B chunks XOR-swizzled by token&15) -> rows ------
    // 32x32 accumulator: lane = token column (l&31); register i = weight row (i&3) + 8*(i>>2) + 4*(l>>5) of the tile
    __syncthreads();   // every wave is done reading the ring
#ifdef SGLK_DEV_ABLATE
    if (p.dbg && tid == 0) p.dbg[32 * blockIdx.x + 25] = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int kCols = (MODE == MODE_GATE_UP) ? 128 : 256;   // output columns per workgroup
    constexpr int kRowB = kCols * 2;                            // bytes per token row in the image
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        if (!active) break;
        const int r = wm * 128 + tt * 32 + r32;
        unsigned char* rowp = smem + r * kRowB;
        if (MODE == MODE_GATE_UP) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float g4[4], u4[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { g4[i] = acc[0][tt][rg * 4 + i] * mant[0]; u4[i] = acc[1][tt][rg * 4 + i] * mant[1]; }
                uint2 v;
                v.x = pack_bf16x2(silu_f32(g4[0]) * u4[0], silu_f32(g4[1]) * u4[1]);
                v.y = pack_bf16x2(silu_f32(g4[2]) * u4[2], silu_f32(g4[3]) * u4[3]);
                const int col = wn * 32 + rg * 8 + h * 4;            // 4 consecutive ic1 columns
                const int chunk = (col >> 3) ^ (r & 15);
                *reinterpret_cast<uint2*>(rowp + chunk * 16 + (col & 4) * 2) = v;
            }
        } else if (RESCALE & 4096) {   // variant: registers -> global directly (no LDS image, nothing to wait for)
            if (r < rows) {
                const int slot = slot_tab[r];
                const float tw = tw_tab[r];
                uint16_t* orow = p.out + (int64_t)slot * p.out_stride + ntile * kCols + wn * 64 + h * 4;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const float sc_w = mant[rt] * tw;
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        uint2 v;
                        v.x = pack_bf16x2(acc[rt][tt][rg * 4 + 0] * sc_w, acc[rt][tt][rg * 4 + 1] * sc_w);
                        v.y = pack_bf16x2(acc[rt][tt][rg * 4 + 2] * sc_w, acc[rt][tt][rg * 4 + 3] * sc_w);
                        *reinterpret_cast<uint2*>(orow + rt * 32 + rg * 8) = v;
                    }
                }
            }
        } else {
            float tw = 1.f;
            if (MODE == MODE_DOWN) tw = tw_tab[r];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const float sc_w = mant[rt] * tw;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    float o4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) o4[i] = acc[rt][tt][rg * 4 + i] * sc_w;
                    if (MODE == MODE_PLAIN) {   // dense epilogue: + bias[col] + addend[row][col] * scale, in fp32
                        const int gc = ntile * kCols + wn * 64 + rt * 32 + rg * 8 + h * 4;
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
                    const int col = wn * 64 + rt * 32 + rg * 8 + h * 4;
                    const int chunk = (col >> 3) ^ (r & 15);
                    *reinterpret_cast<uint2*>(rowp + chunk * 16 + (col & 4) * 2) = v;
                }
            }
        }
    }
    if (MODE == MODE_DOWN && (RESCALE & 4096)) return;
#ifdef SGLK_DEV_ABLATE
    if (p.dbg && tid == 0) p.dbg[32 * blockIdx.x + 26] = __builtin_amdgcn_s_memrealtime();
#endif
    __syncthreads();
#ifdef SGLK_DEV_ABLATE
    if (p.dbg && tid == 0) p.dbg[32 * blockIdx.x + 27] = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int kChunksPerRow = kRowB / 16;                   // 16 or 32
    constexpr int kIters = kBM * kChunksPerRow / 512;           // 8 or 16
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        const int idx = it * 512 + tid;
        const int r = idx / kChunksPerRow;
        const int pc = idx - r * kChunksPerRow;                 // physical chunk
        const int lc = pc ^ (r & 15);                           // logical chunk = 8 columns
        if (r < rows) {
            const uint4 v = *reinterpret_cast<const uint4*>(smem + r * kRowB + pc * 16);
            int64_t orow;
            if (MODE == MODE_DOWN) orow = (int64_t)slot_tab[r] * p.out_stride + ntile * kCols;
            else orow = (int64_t)(pos0 + r) * p.out_stride + ntile * kCols;
            *reinterpret_cast<uint4*>(p.out + orow + lc * 8) = v;
        }
    }
#ifdef SGLK_DEV_ABLATE
    if (p.dbg && tid == 0) {
        p.dbg[32 * blockIdx.x + 21] = __builtin_amdgcn_s_memrealtime();   // stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p.dbg[32 * blockIdx.x + 24] = __builtin_amdgcn_s_memrealtime();   // stores acknowledged
    }
#endif
}

}  // namespace g256x

int launch_moe_gemm_fp8w_256x(int mode, const MoeGemmParams& p, int max_mtiles, hipStream_t stream) {
    const int64_t blocks = (int64_t)max_mtiles * p.n_tiles;
    if (blocks == 0) return SGLK_OK;
    if ((p.C >> 7) > g256x::kMaxKBlocks) SGLK_FAIL(SGLK_ERR_SHAPE, "moe_gemm_fp8w_256x: reduction length %d too long", p.C);
    static const bool keep_x = getenv("SGLK_G256X") != nullptr;
    if (!keep_x && (p.C >> 7) >= 2) return launch_moe_gemm_fp8w_256i(mode, p, max_mtiles, stream);
    if (mode == MODE_PLAIN) {
        hipLaunchKernelGGL((g256x::moe_gemm_fp8w_256x_kernel<MODE_PLAIN, 0>), dim3((unsigned)blocks), dim3(512), 0, stream, p);
        SGLK_CHECK_LAUNCH("moe_gemm_fp8w_256x");
        return SGLK_OK;
    }
#define SGLK_LAUNCH256X(R)                                                                                             \
    if (mode == MODE_GATE_UP)                                                                                          \
        hipLaunchKernelGGL((g256x::moe_gemm_fp8w_256x_kernel<MODE_GATE_UP, R>), dim3((unsigned)blocks), dim3(512), 0, stream, p); \
    else                                                                                                               \
        hipLaunchKernelGGL((g256x::moe_gemm_fp8w_256x_kernel<MODE_DOWN, R>), dim3((unsigned)blocks), dim3(512), 0, stream, p)
#ifdef SGLK_DEV_ABLATE   // developer-only timing ablations (wrong results by design)
    static const int abl = getenv("SGLK_RESCALE") ? atoi(getenv("SGLK_RESCALE")) : 0;
    switch (abl) {
        case 4: SGLK_LAUNCH256X(4); break;
        case 28: SGLK_LAUNCH256X(28); break;
        case 64: SGLK_LAUNCH256X(64); break;
        case 92: SGLK_LAUNCH256X(92); break;
        case 256: SGLK_LAUNCH256X(256); break;
        case 1280: SGLK_LAUNCH256X(1280); break;
        case 4096: SGLK_LAUNCH256X(4096); break;
        case 2048: SGLK_LAUNCH256X(2048); break;
        case 512: SGLK_LAUNCH256X(512); break;
        default: SGLK_LAUNCH256X(0); break;
    }
#else
    SGLK_LAUNCH256X(0);
#endif
#undef SGLK_LAUNCH256X
    SGLK_CHECK_LAUNCH("moe_gemm_fp8w_256x");
    return SGLK_OK;
}

}  // namespace sglk
