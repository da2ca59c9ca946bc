// Dense W8A8 GEMM on the int8 matrix cores:  out[m][n] = (sum_k xq[m][k] * wq[n][k]) * xs[m] * ws[n] (+ bias[n]).
//
// Replaces the large-M case of torch.ops.sgl_kernel.int8_scaled_mm_cpu and int8_scaled_mm_with_quant
// (/root/reference/test_gemm_int8.py:67,72); oracle native_w8a8_per_token_matmul (test_gemm_int8.py:24-47:
// C = As * (A . B^T) * Bs + bias, fp32, then the output cast).  The integer dot products are accumulated EXACTLY in
// int32 by mfma_i32_32x32x32_i8 (twice the bf16 MFMA rate), the two scales and the bias are applied in fp32 in the
// oracle's order, one rounding to bf16.
//
// The same kernel serves the int8 fused_experts (/root/reference/test_moe_int8.py:59-94, bench_moe.py:89-106) as its two
// grouped GEMMs: MODE_GATE_UP gathers the per-token-quantised activations through sorted_slot and writes
// silu(gate) * up in fp32 (the per-row re-quantisation needs the whole row, so it stays a separate pass), MODE_DOWN
// applies the routing weight and scatters bf16 rows by slot.
//
// Tile 256 tokens x 256 weight rows per workgroup, 8 waves (4 along weight rows x 2 along tokens), wave tile 64 x 128 =
// 2 x 4 MFMA tiles, 128 int32 accumulators per lane, 2 waves per SIMD.  K in 64-deep stages (two k-steps of 32) through
// a ring of FOUR LDS buffers (X 16 KiB + W 16 KiB each): both operands arrive by LDS-DMA three stages ahead, one counted
// s_waitcnt vmcnt + s_barrier per stage.  No conversion, no rescale: the feed of the next k-step (2 weight + 4
// activation ds_read_b128, the DMA pieces) is issued MFMA by MFMA in the shadows (moe_gemm_fp8w_256i.hip).
//   * packed int8 weight tile (pack.hip): 16 rows x 64 k = 1 KiB, lane (r = l&15, g = l>>4) holds k = 16g .. 16g+15 of
//     row r -> the A operand of lane (row r32, k half h) at k-step ks is lane slot (2ks+h)*16 + (r32&15) of piece
//     (r32>>4): one conflict-free ds_read_b128;
//   * activations: LDS image [256 rows][64 B], 16-byte chunks XOR-swizzled by (row>>2)&3 (applied to the DMA source
//     address), so the B-operand ds_read_b128 of every lane group is conflict-free.
#include "sglk_common.h"
#include "moe_internal.h"

// no mul+add contraction in this file: the epilogue reproduces the oracle's separately rounded fp32 operations
#pragma clang fp contract(off)

namespace sglk {

typedef __attribute__((address_space(3))) void* lptr_t;

namespace gi8 {

typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4v;

constexpr int kBM = 256;
constexpr int kStageX = kBM * 64;          // 16 KiB
constexpr int kStageW = 16 * 1024;         // 16 packed 16x64 tiles
constexpr int kStage = kStageX + kStageW;  // 32 KiB
constexpr int kRing = 4;
constexpr int kTabOff = kRing * kStage;    // 128 KiB, then xs[256], ws[256], bias-or-routing-weight[256] (f32), slot[256]
constexpr int kLds = kTabOff + 4 * 256 * 4;

template <int MODE>
__global__ __launch_bounds__(512, 2) void gemm_i8_256_kernel(const I8GemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[kLds];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wm = wave >> 2;

    int live, e = 0, pos0, rows, mtile, ntile, ksr = 0;
    // dense split-K (launches that would not fill the chip): workgroup = (m-tile, column tile, K range), exact int32 partials
    const bool ksp = MODE == MODE_PLAIN && !p.tile_info && p.ksplit > 1;
    if (p.tile_info) {   // grouped: m-tile table of moe_align (tile_m = 256)
        live = p.num_tiles[0] * p.n_tiles;
        if ((int)blockIdx.x >= live) return;
        const int L = xcd_remap(blockIdx.x, live);
        mtile = L / p.n_tiles;
        ntile = L - mtile * p.n_tiles;
        const int4 ti = p.tile_info[mtile];
        e = __builtin_amdgcn_readfirstlane(ti.x);
        pos0 = __builtin_amdgcn_readfirstlane(ti.y);
        rows = __builtin_amdgcn_readfirstlane(ti.z);
    } else {             // dense: rows in natural order
        const int mtiles = (p.M + kBM - 1) / kBM;
        const int nks = ksp ? p.ksplit : 1;
        live = mtiles * p.n_tiles * nks;
        if ((int)blockIdx.x >= live) return;
        // consecutive workgroups of an XCD share the activation rows (same m-tile) and walk the weight row tiles
        const int L0 = xcd_remap(blockIdx.x, live);
        ksr = L0 % nks;
        const int L = L0 / nks;
        mtile = L / p.n_tiles;
        ntile = L - mtile * p.n_tiles;
        pos0 = mtile * kBM;
        rows = (p.M - pos0 < kBM) ? p.M - pos0 : kBM;
    }
    // workgroup's 16 weight row-tiles: GATE_UP = 8 gate + 8 up (the same 128 ic1 columns), else 16 consecutive
    auto piece_row16 = [&](int piece) {
        if (MODE == MODE_GATE_UP) return (piece < 8) ? ntile * 8 + piece : (p.n_half >> 4) + ntile * 8 + (piece - 8);
        return ntile * 16 + piece;
    };
    const int T = p.K >> 6;   // 64-deep stages, >= 4 (launcher)

    float* xs_tab = reinterpret_cast<float*>(smem + kTabOff);
    float* ws_tab = xs_tab + 256;
    float* bias_tab = ws_tab + 256;                       // PLAIN: bias per column; DOWN: routing weight per row
    int* slot_tab = reinterpret_cast<int*>(bias_tab + 256);   // DOWN: output row (slot) per tile row
    // per-row / per-column epilogue factors: fetched now, parked in registers, written to LDS after the DMA is out
    float my_a = 0.f, my_b = 0.f;
    int my_slot = -1;
    int64_t my_xrow = 0;      // x row of tile row `tid` (tid < 256)
    if (tid < kBM) {
        if (tid < rows) {
            if (MODE == MODE_GATE_UP) {
                my_slot = p.sorted_slot[pos0 + tid];
                my_xrow = my_slot / p.topk;
            } else {
                my_xrow = pos0 + tid;
                if (MODE == MODE_DOWN) my_slot = p.sorted_slot[pos0 + tid];
            }
            my_a = p.x_scale[my_xrow];
            if (MODE == MODE_DOWN) my_b = p.topk_weights[my_slot];
        }
    } else {
        const int c = tid - 256;
        const float* sc_e = p.w_scale + (int64_t)e * p.scale_rows;
        if (MODE == MODE_GATE_UP) my_a = sc_e[c < 128 ? ntile * 128 + c : p.n_half + ntile * 128 + (c - 128)];
        else my_a = sc_e[ntile * 256 + c];
        if (MODE == MODE_PLAIN && p.bias) my_b = p.bias[ntile * 256 + c];
    }

    // ---- LDS-DMA sources ------------------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (int64_t)e * p.w_bytes), 0, (unsigned)p.w_bytes, 0x00020000);
    const int ctiles = (ksp ? p.split_kblocks * p.ksplit * 2 : p.K >> 6);   // 64-wide tiles per weight row (whole reduction)
    const unsigned x_koff = ksp ? (unsigned)ksr * (unsigned)p.K : 0u;                 // bytes into every row of x
    const unsigned w_koff = ksp ? (unsigned)ksr * (unsigned)(p.K >> 6) * 1024u : 0u;  // this range's first tile of a row tile
    // X piece i of the wave (i = 0,1): image rows (wave*2+i)*16 + (lane>>2), LDS chunk lane&3 <- source chunk ^ swizzle
    unsigned xsrc[2], wsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 16 + (lane >> 2);
        const int rr = r < rows ? r : 0;
        int64_t xrow = pos0 + rr;
        if (MODE == MODE_GATE_UP) xrow = p.sorted_slot[pos0 + rr] / p.topk;
        xsrc[i] = (unsigned)(xrow * p.x_stride) + x_koff + (unsigned)((((lane & 3) ^ ((r >> 2) & 3))) << 4);
        wsrc[i] = (unsigned)(piece_row16(wave * 2 + i) * ctiles) * 1024u + w_koff + lane * 16;
    }
    auto issue_piece = [&](int kt, int buf, int i) __attribute__((always_inline)) {   // i = 0,1: X rows; 2,3: packed W tiles
        unsigned char* sx = smem + buf * kStage;
        if (i < 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lptr_t)(sx + (wave * 2 + i) * 1024), 16, xsrc[i], kt * 64, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lptr_t)(sx + kStageX + (wave * 2 + i - 2) * 1024), 16,
                                                     wsrc[i - 2], kt * 1024, 0, 0);
    };

    // ---- operand addressing (A = weights, B = tokens) --------------------------------------------------------------------
    const int h = lane >> 5, r32 = lane & 31;
    int wbase[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        // GATE_UP: row tile 0 = gate pieces wn*2.., row tile 1 = the matching up pieces (same columns, same lane)
        const int piece0 = (MODE == MODE_GATE_UP) ? (rt == 0 ? wn * 2 : 8 + wn * 2) : wn * 4 + rt * 2;
        wbase[rt] = (piece0 + (r32 >> 4)) * 1024 + (r32 & 15) * 16;
    }
    auto woff = [&](int rt, int ks) __attribute__((always_inline)) { return wbase[rt] + (2 * ks + h) * 256; };
    const int xrow0 = wm * 128 + r32;
    auto xoff = [&](int tt, int ks) __attribute__((always_inline)) {
        const int row = xrow0 + tt * 32;
        return row * 64 + (((2 * ks + h) ^ ((row >> 2) & 3)) << 4);
    };

    i32x16 acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][tt][i] = 0;

    // prologue: up to four stages in flight
#pragma unroll
    for (int st = 0; st < kRing; ++st) {
        if (st < T) {
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_piece(st, st, i);
        }
    }
    if (tid < kBM) {
        xs_tab[tid] = my_a;
        if (MODE == MODE_DOWN) { bias_tab[tid] = my_b; slot_tab[tid] = my_slot; }
    } else {
        ws_tab[tid - 256] = my_a;
        if (MODE == MODE_PLAIN) bias_tab[tid - 256] = my_b;
    }
    // stage 0 landed (its four pieces are the oldest of up to 16), tables visible
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // T >= 4
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

#define SGLK_FENCE() __builtin_amdgcn_sched_barrier(0)
    i32x4v wf[2][2], xf[2][4];   // [k-step parity][row tile / token tile]
    auto ld_w = [&](int par, int rt, int fbuf, int ks) __attribute__((always_inline)) {
        wf[par][rt] = *reinterpret_cast<const i32x4v*>(smem + fbuf * kStage + kStageX + woff(rt, ks));
    };
    auto ld_x = [&](int par, int tt, int fbuf, int ks) __attribute__((always_inline)) {
        xf[par][tt] = *reinterpret_cast<const i32x4v*>(smem + fbuf * kStage + xoff(tt, ks));
    };
    auto mma = [&](int par, int s2) __attribute__((always_inline)) {
        const int rt = (s2 >> 1) & 1, tt = (s2 & 1) + 2 * (s2 >> 2);
        acc[rt][tt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[par][rt], xf[par][tt], acc[rt][tt], 0, 0, 0);
    };

    const bool active = wm * 128 < rows;
    int buf = 0;
    // k-step ks (0/1) of the stage in `buf`.  fetch: read the next k-step's operands (k-step 1 of this stage, or
    // k-step 0 of the next one after the sync).  In k-step 1 the sync point S_t comes after the second MFMA: the next
    // stage has landed and every wave is done with this stage's buffer, which then receives stage t+4.
    auto kstep = [&](int t, int ks, bool fetch, bool sync, int wait_pieces, bool refill) __attribute__((always_inline)) {
        const int par = ks, npar = ks ^ 1;
        const int nbuf = (buf + 1) & (kRing - 1);
        const int fbuf = ks ? nbuf : buf;
        const int fks = ks ^ 1;
        mma(par, 0);
        SGLK_FENCE();
        if (sync) {
            mma(par, 1);
            SGLK_FENCE();
            if (wait_pieces == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (wait_pieces == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (fetch) { ld_w(npar, 0, fbuf, fks); ld_w(npar, 1, fbuf, fks); ld_x(npar, 0, fbuf, fks); ld_x(npar, 1, fbuf, fks); }
            SGLK_FENCE();
        } else {
            if (fetch) { ld_w(npar, 0, fbuf, fks); ld_x(npar, 0, fbuf, fks); }
            SGLK_FENCE();
            mma(par, 1);
            SGLK_FENCE();
            if (fetch) { ld_w(npar, 1, fbuf, fks); ld_x(npar, 1, fbuf, fks); }
            SGLK_FENCE();
        }
        mma(par, 2);
        SGLK_FENCE();
        if (fetch) ld_x(npar, 2, fbuf, fks);
        SGLK_FENCE();
        mma(par, 3);
        SGLK_FENCE();
        if (fetch) ld_x(npar, 3, fbuf, fks);
        if (refill) issue_piece(t + 4, buf, 0);
        SGLK_FENCE();
        mma(par, 4);
        SGLK_FENCE();
        if (refill) issue_piece(t + 4, buf, 1);
        SGLK_FENCE();
        mma(par, 5);
        SGLK_FENCE();
        mma(par, 6);
        SGLK_FENCE();
        if (refill) issue_piece(t + 4, buf, 2);
        SGLK_FENCE();
        mma(par, 7);
        SGLK_FENCE();
        if (refill) issue_piece(t + 4, buf, 3);
        SGLK_FENCE();
    };
    auto idle_stage = [&](int t, bool has_next, int wait_pieces, bool refill) __attribute__((always_inline)) {
        if (has_next) {
            if (wait_pieces == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (wait_pieces == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (refill) {
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_piece(t + 4, buf, i);
        }
        buf = (buf + 1) & (kRing - 1);
    };

    if (active) {
        ld_w(0, 0, 0, 0);
        ld_w(0, 1, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) ld_x(0, tt, 0, 0);
        SGLK_FENCE();
        int t = 0;
        for (; t + 4 < T; ++t) {            // steady state: stage t+4 exists
            kstep(t, 0, true, false, 0, false);
            kstep(t, 1, true, true, 8, true);
            buf = (buf + 1) & (kRing - 1);
        }
        // drain (T >= 4, guaranteed by the launcher): stages T-4 .. T-1 without refill; the waits are literals so that
        // no control flow (and no register spill, whose scratch traffic would corrupt the vmcnt accounting) appears
        kstep(t, 0, true, false, 0, false);
        kstep(t, 1, true, true, 8, false);      // T-4: stages T-2, T-1 may still be in flight
        buf = (buf + 1) & (kRing - 1);
        ++t;
        kstep(t, 0, true, false, 0, false);
        kstep(t, 1, true, true, 4, false);      // T-3: stage T-1 may still be in flight
        buf = (buf + 1) & (kRing - 1);
        ++t;
        kstep(t, 0, true, false, 0, false);
        kstep(t, 1, true, true, 0, false);      // T-2: everything must have landed
        buf = (buf + 1) & (kRing - 1);
        ++t;
        kstep(t, 0, true, false, 0, false);     // last stage: its second k-step fetches nothing
        kstep(t, 1, false, false, 0, false);
    } else {
        int t = 0;
        for (; t + 4 < T; ++t) idle_stage(t, true, 8, true);
        idle_stage(t, true, 8, false);
        idle_stage(t + 1, true, 4, false);
        idle_stage(t + 2, true, 0, false);
    }
#undef SGLK_FENCE

    // ---- epilogue: int32 -> fp32, scales in the oracle's order (As * C * Bs), image in LDS, whole rows out ---------------
    // image rows are 512 B either way: 256 bf16 columns (PLAIN / DOWN) or 128 fp32 columns (GATE_UP)
    __syncthreads();   // every wave is done reading the ring
    constexpr int kRowB = 512;
    if (active) {
        float xs4[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) xs4[tt] = xs_tab[wm * 128 + tt * 32 + r32];
        if (MODE == MODE_GATE_UP) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int col = wn * 32 + rg * 8 + h * 4;            // 4 consecutive ic1 columns
                const float4 wg = *reinterpret_cast<const float4*>(ws_tab + col);
                const float4 wu = *reinterpret_cast<const float4*>(ws_tab + 128 + col);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int r = wm * 128 + tt * 32 + r32;
                    const float xs = xs4[tt];
                    float4 v;
                    v.x = silu_f32(xs * (float)acc[0][tt][rg * 4 + 0] * wg.x) * (xs * (float)acc[1][tt][rg * 4 + 0] * wu.x);
                    v.y = silu_f32(xs * (float)acc[0][tt][rg * 4 + 1] * wg.y) * (xs * (float)acc[1][tt][rg * 4 + 1] * wu.y);
                    v.z = silu_f32(xs * (float)acc[0][tt][rg * 4 + 2] * wg.z) * (xs * (float)acc[1][tt][rg * 4 + 2] * wu.z);
                    v.w = silu_f32(xs * (float)acc[0][tt][rg * 4 + 3] * wg.w) * (xs * (float)acc[1][tt][rg * 4 + 3] * wu.w);
                    const int chunk = (col >> 2) ^ (r & 15);         // 16-byte chunk = 4 fp32 columns
                    *reinterpret_cast<float4*>(smem + r * kRowB + chunk * 16) = v;
                }
            }
        } else {
            float tw4[4] = {1.f, 1.f, 1.f, 1.f};
            if (MODE == MODE_DOWN) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) tw4[tt] = bias_tab[wm * 128 + tt * 32 + r32];
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    // the column factors are fetched once per (row tile, register group) and used for the four token tiles
                    const int col = wn * 64 + rt * 32 + rg * 8 + h * 4;
                    const float4 w4 = *reinterpret_cast<const float4*>(ws_tab + col);
                    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (MODE == MODE_PLAIN) b4 = *reinterpret_cast<const float4*>(bias_tab + col);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const int r = wm * 128 + tt * 32 + r32;
                        if (ksp) {   // this range's exact int32 sums; scales and bias are applied by the reduce
                            if (r < rows)
                                *reinterpret_cast<int4*>(p.partial_i32 + ((int64_t)ksr * p.M + pos0 + r) * p.N + ntile * 256 + col) =
                                    make_int4(acc[rt][tt][rg * 4 + 0], acc[rt][tt][rg * 4 + 1], acc[rt][tt][rg * 4 + 2], acc[rt][tt][rg * 4 + 3]);
                            continue;
                        }
                        const float xs = xs4[tt];
                        float o4[4];   // (xs * acc) * ws (+ bias | * routing weight), separately rounded like the oracle
                        o4[0] = xs * (float)acc[rt][tt][rg * 4 + 0] * w4.x;
                        o4[1] = xs * (float)acc[rt][tt][rg * 4 + 1] * w4.y;
                        o4[2] = xs * (float)acc[rt][tt][rg * 4 + 2] * w4.z;
                        o4[3] = xs * (float)acc[rt][tt][rg * 4 + 3] * w4.w;
                        if (MODE == MODE_PLAIN) { o4[0] += b4.x; o4[1] += b4.y; o4[2] += b4.z; o4[3] += b4.w; }
                        if (MODE == MODE_PLAIN && p.addend && r < rows) {   // + fused_out * routed_scaling_factor (shared expert)
                            const uint2 av = *reinterpret_cast<const uint2*>(p.addend + (int64_t)(pos0 + r) * p.addend_stride + ntile * 256 + col);
                            o4[0] += __uint_as_float(av.x << 16) * p.addend_scale;
                            o4[1] += __uint_as_float(av.x & 0xffff0000u) * p.addend_scale;
                            o4[2] += __uint_as_float(av.y << 16) * p.addend_scale;
                            o4[3] += __uint_as_float(av.y & 0xffff0000u) * p.addend_scale;
                        }
                        if (MODE == MODE_DOWN) { o4[0] *= tw4[tt]; o4[1] *= tw4[tt]; o4[2] *= tw4[tt]; o4[3] *= tw4[tt]; }
                        uint2 v;
                        v.x = pack_bf16x2(o4[0], o4[1]);
                        v.y = pack_bf16x2(o4[2], o4[3]);
                        const int chunk = (col >> 3) ^ (r & 15);
                        *reinterpret_cast<uint2*>(smem + r * kRowB + chunk * 16 + (col & 4) * 2) = v;
                    }
                }
            }
        }
    }
    __syncthreads();
    constexpr int kChunksPerRow = kRowB / 16;          // 32
    constexpr int kIters = kBM * kChunksPerRow / 512;  // 16
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        const int idx = it * 512 + tid;
        const int r = idx / kChunksPerRow;
        const int pc = idx - r * kChunksPerRow;
        const int lc = pc ^ (r & 15);                  // logical chunk: 8 bf16 or 4 fp32 columns
        if (r < rows && !ksp) {
            const uint4 v = *reinterpret_cast<const uint4*>(smem + r * kRowB + pc * 16);
            if (MODE == MODE_GATE_UP) {
                float* orow = reinterpret_cast<float*>(p.out) + (int64_t)(pos0 + r) * p.out_stride + ntile * 128;
                *reinterpret_cast<uint4*>(orow + lc * 4) = v;
            } else {
                const int64_t orow = (MODE == MODE_DOWN) ? (int64_t)slot_tab[r] : (int64_t)(pos0 + r);
                *reinterpret_cast<uint4*>(p.out + orow * p.out_stride + ntile * 256 + lc * 8) = v;
            }
        }
    }
}

}  // namespace gi8

int launch_gemm_i8_256(int mode, const I8GemmParams& p, int max_mtiles, hipStream_t stream) {
    if (p.K < 256 || p.K % 64 != 0) SGLK_FAIL(SGLK_ERR_SHAPE, "gemm_i8_256: reduction length %d must be a multiple of 64 and >= 256", p.K);
    const int64_t blocks = (int64_t)max_mtiles * p.n_tiles * ((mode == MODE_PLAIN && !p.tile_info && p.ksplit > 1) ? p.ksplit : 1);
    if (blocks <= 0) return SGLK_OK;
    if (mode == MODE_GATE_UP) hipLaunchKernelGGL(gi8::gemm_i8_256_kernel<MODE_GATE_UP>, dim3((unsigned)blocks), dim3(512), 0, stream, p);
    else if (mode == MODE_DOWN) hipLaunchKernelGGL(gi8::gemm_i8_256_kernel<MODE_DOWN>, dim3((unsigned)blocks), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL(gi8::gemm_i8_256_kernel<MODE_PLAIN>, dim3((unsigned)blocks), dim3(512), 0, stream, p);
    SGLK_CHECK_LAUNCH("gemm_i8_256");
    return SGLK_OK;
}

}  // namespace sglk
