// Decode-size dense bf16 GEMM (weight_packed_linear with the reference's VNNI-2 packed weights, M < 192): the
// weight-streaming scheme of moe_gemm_fp8w_mid.hip without the conversion.
//
// out[M][N] = x[M][K] . w[N][K]^T (+ bias), oracle /root/reference/test_gemm.py:15-25.  BASELINE.json config 0 is the
// (128, 4096, 4096) case.
//
// * workgroup = up to 64 (M <= 64) or 128 rows of x  x  128 weight rows  x  one K range; 8 waves, each owns one 16-row weight tile for all
//   the rows; two workgroups per CU (<= 128 VGPRs) so that one's prologue hides behind the other's stream;
// * weights never touch LDS.  The packed order [N/32][K/2][32 rows][2 k] keeps, for one k pair, the 32 rows' dwords
//   together: lane (row r, k group g) gathers the four k pairs of its MFMA A-operand octet with four global_load_dword
//   (128 B apart), i.e. a 128-wide K block of the wave's tile is 16 dword loads per lane; two blocks are in flight;
// * x goes global -> LDS by DMA one K block at a time, double buffered (swizzle applied to the source address), one
//   counted s_waitcnt + one barrier per block -- waits are builtins so that the compiler's own wait-count pass sees them;
// * the reduction is cut into equal ranges of an even number (>= 4) of K blocks until ~512 workgroups exist; fp32 partials
//   [range][row][N] are summed in range order by the generic engine's reduce (bias and the bf16 rounding happen there).
#include "knobs.h"
#include "moe_internal.h"

namespace sglk {

typedef const __attribute__((address_space(1))) void* gptr_bm_t;
typedef __attribute__((address_space(3))) void* lptr_bm_t;

namespace gbmid {

// Two builds: TM = 64 rows per tile at most (<= 128 VGPRs, two workgroups per CU: one's prologue hides behind the other's
// stream) for M <= 64, and TM = 128 (one workgroup per CU) for larger M -- with 64-row tiles a 128-row problem streams the
// weights twice (FETCH_SIZE 73 MB for the 32 MiB matrix of the (128, 4096, 4096) case) and is bound by exactly that.

constexpr int vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }   // s_waitcnt vmcnt(n) only (gfx9 encoding)

struct Ctx {
    int pos0, rows, ntile, kblocks, kb0, ksr;
    const unsigned char* wp[2];   // lane's dword of (tile row, k pair 4g) in K block 0 of the range, per weight tile
};

// MODE: PLAIN = dense rows (split-K, bias); GATE_UP / DOWN = the two grouped GEMMs of bf16 fused_experts at small / mid batch
// sizes (gather by sorted_slot, gate + up tile per wave with SiLU*mul in registers; routing weight + scatter by slot), same
// contract as gemm_bf16_256.hip's modes (oracle /root/reference/test_moe.py:22-54).
// NW = waves per workgroup (8; 4 for GATE_UP launches that reach at most half the CUs), XD = K blocks the activations travel ahead
// (1; 3 with four LDS buffers of the 32 rows a decode-size tile has): as in moe_gemm_fp8w_mid.hip, where the two are measured
template <int MODE, int MT, int TM, bool ODD, int NW = 8, int XD = 1>
SGLK_DEV void run(const BmidParams& p, unsigned char* lds, const Ctx& c) {
    static_assert(NW == 8 || MODE == MODE_GATE_UP, "narrow workgroups exist for GATE_UP only");
    static_assert(XD == 1 || (XD == 3 && MT == 2 && !ODD && MODE == MODE_GATE_UP), "far prefetch: short GATE_UP tiles, even block counts");
    constexpr int kXBuf = TM * 256;
    constexpr int XB = XD + 1;
    constexpr int kXSz = XD == 1 ? kXBuf : MT * 16 * 256;
    constexpr int TPW = MODE == MODE_GATE_UP ? 2 : 1;
    // one K block of the wave's tile = 4 k-steps x 4 k pairs; k pair kp of the block sits kp * 128 B further
    auto load_block = [&](u32x4 (&dst)[TPW][4], int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < TPW; ++a) {
            const unsigned char* b = c.wp[a] + (int64_t)kb * (64 * 128);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[a][s][j] = *reinterpret_cast<const unsigned*>(b + (s * 16 + j) * 128);
        }
    };
    // DOWN: output slot and routing weight of the tile's rows start their two round trips here and wait in LDS
    int my_slot = -1;
    float my_tw = 0.f;
    if (MODE == MODE_DOWN && (int)threadIdx.x < c.rows) my_slot = p.sorted_slot[c.pos0 + threadIdx.x];
    u32x4 ring[2][TPW][4];
    load_block(ring[0], 0);
    load_block(ring[1], 1);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    constexpr int XV = MT * 4 / NW;

    const uint16_t* xsrc[XV];
#pragma unroll
    for (int j = 0; j < XV; ++j) {
        const int row = (wave * XV + j) * 4 + (lane >> 4), ch = (lane & 15) ^ (row & 15);
        const int rr = row < c.rows ? row : c.rows - 1;     // padding rows re-read the last row; their outputs are dropped
        int64_t xrow;
        if (MODE == MODE_GATE_UP) xrow = (int64_t)(p.sorted_slot[c.pos0 + rr] / p.topk) * p.x_stride;
        else xrow = (int64_t)(c.pos0 + rr) * p.x_stride;
        xsrc[j] = p.x + xrow + ch * 8 + (int64_t)c.kb0 * 128;
    }
    auto x_dma = [&](int kb) __attribute__((always_inline)) {
        unsigned char* dst = lds + (kb % XB) * kXSz + wave * XV * 1024;
#pragma unroll
        for (int j = 0; j < XV; ++j)
            __builtin_amdgcn_global_load_lds((gptr_bm_t)(xsrc[j] + kb * 128), (lptr_bm_t)(dst + j * 1024), 16, 0, 0);
    };

    f32x4 acc[TPW][MT];
#pragma unroll
    for (int a = 0; a < TPW; ++a)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[a][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // has_x: block kb+XD exists (requested here); prefetch_x: block kb+1 exists (sync at the end of this block)
    auto block = [&](int kb, int half, bool refill, bool prefetch_x, bool has_x) __attribute__((always_inline)) {
        if (has_x) x_dma(kb + XD);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 w[TPW][4];
#pragma unroll
        for (int a = 0; a < TPW; ++a)
#pragma unroll
            for (int s = 0; s < 4; ++s) w[a][s] = __builtin_bit_cast(bf16x8, ring[half][a][s]);
        if (refill) load_block(ring[half], kb + 2);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* xb = lds + (kb % XB) * kXSz;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int xr = mt * 16 + r;
            const unsigned char* base = xb + xr * 256;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 x = *reinterpret_cast<const bf16x8*>(base + (((s * 4 + g) ^ (xr & 15)) << 4));
#pragma unroll
                for (int a = 0; a < TPW; ++a) acc[a][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[a][s], x, acc[a][mt], 0, 0, 0);
            }
        }
        if (prefetch_x) {
            if (XD == 3 && has_x && refill) __builtin_amdgcn_s_waitcnt(vmcnt_imm(XV + 16 * TPW));   // this block's own requests stay in flight
            else if (refill) __builtin_amdgcn_s_waitcnt(vmcnt_imm(16 * TPW));   // all but the refill loads: the DMA has landed
            else __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
            __builtin_amdgcn_s_barrier();
        }
    };

    if (MODE == MODE_DOWN && my_slot >= 0) my_tw = p.topk_weights[my_slot];
    x_dma(0);
    if (XD == 3) {
        x_dma(1);
        x_dma(2);
    }
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
    int* slot_tab = reinterpret_cast<int*>(lds + 2 * kXBuf);
    float* tw_tab = reinterpret_cast<float*>(lds + 2 * kXBuf + TM * 4);
    if (MODE == MODE_DOWN && tid < TM) {
        slot_tab[tid] = my_slot;
        tw_tab[tid] = my_tw;
    }
    __syncthreads();
    int kb = 0;
    if (XD == 3) {   // even count >= 4: pairs with everything on, then the last four blocks with literal flags
        for (; kb + 5 <= c.kblocks; kb += 2) {
            block(kb, 0, true, true, true);
            block(kb + 1, 1, true, true, true);
        }
        block(kb, 0, true, true, true);
        block(kb + 1, 1, true, true, false);
        block(kb + 2, 0, false, true, false);
        block(kb + 3, 1, false, false, false);
    } else if (!ODD) {   // parity of the block count = template parameter (both tails in one kernel cost registers)
        for (; kb + 2 < c.kblocks; kb += 2) {
            block(kb, 0, true, true, true);
            block(kb + 1, 1, true, true, true);
        }
        block(kb, 0, false, true, true);
        block(kb + 1, 1, false, false, false);
    } else {                               // odd block counts (expert widths like 384): a three-block tail
        for (; kb + 3 < c.kblocks; kb += 2) {
            block(kb, 0, true, true, true);
            block(kb + 1, 1, true, true, true);
        }
        block(kb, 0, true, true, true);
        block(kb + 1, 1, false, true, true);
        block(kb + 2, 0, false, false, false);
    }

    const int q4 = g * 4;
    const int col = c.ntile * (NW * 16) + wave * 16 + q4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int tr = mt * 16 + r;
        if (tr >= c.rows) continue;
        if (MODE == MODE_GATE_UP) {
            const f32x4 gt = acc[0][mt], up = acc[TPW - 1][mt];
            uint2 v;
            v.x = pack_bf16x2(silu_f32(gt[0]) * up[0], silu_f32(gt[1]) * up[1]);
            v.y = pack_bf16x2(silu_f32(gt[2]) * up[2], silu_f32(gt[3]) * up[3]);
            *reinterpret_cast<uint2*>(p.out + (int64_t)(c.pos0 + tr) * p.out_stride + col) = v;
            continue;
        }
        if (MODE == MODE_DOWN) {
            const f32x4 v4 = acc[0][mt] * tw_tab[tr];
            uint2 v;
            v.x = pack_bf16x2(v4[0], v4[1]);
            v.y = pack_bf16x2(v4[2], v4[3]);
            *reinterpret_cast<uint2*>(p.out + (int64_t)slot_tab[tr] * p.out_stride + col) = v;
            continue;
        }
        const f32x4 v4 = acc[0][mt];
        if (p.partial) {
            float* dst = p.partial + ((int64_t)c.ksr * p.M + c.pos0 + tr) * p.N + col;
            *reinterpret_cast<float4*>(dst) = make_float4(v4[0], v4[1], v4[2], v4[3]);
        } else {
            float b4[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + col); b4[0] = b.x; b4[1] = b.y; b4[2] = b.z; b4[3] = b.w; }
            uint2 v;
            v.x = pack_bf16x2(v4[0] + b4[0], v4[1] + b4[1]);
            v.y = pack_bf16x2(v4[2] + b4[2], v4[3] + b4[3]);
            *reinterpret_cast<uint2*>(p.out + (int64_t)(c.pos0 + tr) * p.out_stride + col) = v;
        }
    }
}

template <int MODE, int TM, bool ODD, int NW = 8, int XD = 1>
__global__ __launch_bounds__(NW * 64, TM == 64 ? 4 : 2) void gemm_bf16_mid_kernel(const BmidParams p) {
    constexpr int kTM = TM;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nsplit = p.ksplit > 1 ? p.ksplit : 1;
    const int n_tiles = p.N / (NW * 16);
    const bool grouped = MODE != MODE_PLAIN;
    const int mtiles = grouped ? p.num_tiles[0] : (p.M + kTM - 1) / kTM;
    const int live = mtiles * n_tiles * nsplit;
    if ((int)blockIdx.x >= live) return;
    const int Ls = xcd_remap(blockIdx.x, live);
    const int L = Ls / nsplit;
    const int mtile = L / n_tiles;
    Ctx c;
    c.ntile = L - mtile * n_tiles;
    int e = 0;
    if (grouped) {
        const int4 ti = p.tile_info[mtile];
        e = __builtin_amdgcn_readfirstlane(ti.x);
        c.pos0 = __builtin_amdgcn_readfirstlane(ti.y);
        c.rows = __builtin_amdgcn_readfirstlane(ti.z);
    } else {
        c.pos0 = mtile * kTM;
        c.rows = p.M - c.pos0 < kTM ? p.M - c.pos0 : kTM;
    }
    c.ksr = Ls - L * nsplit;
    c.kblocks = nsplit > 1 ? p.split_kblocks : p.K >> 7;
    c.kb0 = c.ksr * c.kblocks;
    // weight row of this lane: tile row (lane & 15) of the wave's 16-row tile; k group g = lane >> 4 -> k pairs 4g .. 4g+3
    const unsigned char* wexp = p.w + (int64_t)e * p.w_expert_stride;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        // GATE_UP: tile 0 = gate rows, tile 1 = the matching up rows (N further); else one tile
        const int R = (a && MODE == MODE_GATE_UP ? p.n_half : 0) + c.ntile * (NW * 16) + wave * 16 + (lane & 15);
        const int64_t dword = ((int64_t)(R >> 5) * (p.K >> 1) + (int64_t)c.kb0 * 64 + (lane >> 4) * 4) * 32 + (R & 31);
        c.wp[a] = wexp + dword * 4;
    }
    const int mt = (c.rows + 15) >> 4;
    constexpr int MTMAX = TM / 16;
    if (mt <= 2) run<MODE, 2, TM, ODD, NW, XD>(p, lds, c);
    else if (MTMAX == 4 || mt <= 4) run<MODE, 4, TM, ODD, NW>(p, lds, c);
    else if (MTMAX == 6 || mt <= 6) run<MODE, (MTMAX >= 6 ? 6 : 4), TM, ODD, NW>(p, lds, c);
    else run<MODE, (MTMAX >= 8 ? 8 : 4), TM, ODD, NW>(p, lds, c);
}

}  // namespace gbmid

// 0 = shape not taken, else the number of K ranges (>= 1); same policy as mid_dense_ksplit
int bf16_mid_ksplit(int M, int N, int K) {
    if (M <= 0 || M >= kMidDenseMaxM || N % 128 != 0 || K % 256 != 0) return 0;   // capability; the dispatch policy is dense_prefers_mid
    const int kblocks = K >> 7;
    const int64_t tiles = (int64_t)ceil_div(M, M <= 64 ? 64 : 128) * (N / 128);
    const int per_min = kblocks >= 4 ? 4 : 2;
    int best = 0;
    if (M > 64 && knobs().bf16_mid_target == 0) {
        // The 128-row build holds ONE workgroup per CU (158-166 VGPRs), so the launch runs ceil(workgroups / CUs) rounds, each as
        // long as one range plus ~3 K blocks' worth of prologue and epilogue: take the split that minimises rounds x (blocks per
        // range + 3).  Aiming at 512 workgroups whatever the rounds (the rule below) ran two rounds of short ranges: 160 x 4096 x
        // 4096 32.5 -> 25.5 us, 512 x 4096 x 4096 48.0 -> 38.9 (tools/ab_bf16_mid_target.py, profiles/r03_ab_dense_129_1000.txt).
        return splitk_by_rounds(kblocks, tiles, M, N, device_cu_count());
    }
    const int target = M > 64 && knobs().bf16_mid_target > 0 ? knobs().bf16_mid_target : 512;
    for (int per = kblocks; per >= per_min; per -= 2) {
        if (kblocks % per != 0) continue;
        const int ks = kblocks / per;
        if (ks > 32 || (int64_t)ks * M * N * 4 > (64ll << 20)) continue;
        best = ks;
        if (tiles * ks >= target) break;
    }
    return best;
}

int launch_gemm_bf16_mid(const BmidParams& p, hipStream_t stream) {
    const int nsplit = p.ksplit > 1 ? p.ksplit : 1;
    const int kblocks = nsplit > 1 ? p.split_kblocks : p.K >> 7;
    if (p.K % 128 != 0 || p.N % 128 != 0 || kblocks < 2 || (nsplit > 1 && (kblocks % 2 != 0 || (p.K >> 7) != nsplit * kblocks || !p.partial)))
        SGLK_FAIL(SGLK_ERR_SHAPE, "gemm_bf16_mid: N=%d K=%d with %d ranges not supported", p.N, p.K, nsplit);
    const int tm = p.M <= 64 ? 64 : 128;
    const int64_t blocks = (int64_t)ceil_div(p.M, tm) * (p.N >> 7) * nsplit;
    if (blocks == 0) return SGLK_OK;
    // split-K ranges are even by construction; an unsplit odd reduction takes the ODD build
    const bool odd = (kblocks & 1) != 0;
#define BMID_PLAIN(TMV, OD)                                                                                        \
    {                                                                                                              \
        const size_t lds = 2 * TMV * 256 + 2 * TMV * 4;                                                            \
        SGLK_ENSURE_DYN_LDS((gbmid::gemm_bf16_mid_kernel<MODE_PLAIN, TMV, OD>), lds, "gemm_bf16_mid");             \
        hipLaunchKernelGGL((gbmid::gemm_bf16_mid_kernel<MODE_PLAIN, TMV, OD>), dim3((unsigned)blocks), dim3(512), lds, stream, p); \
    }
    if (tm == 64) { if (odd) BMID_PLAIN(64, true) else BMID_PLAIN(64, false) }
    else { if (odd) BMID_PLAIN(128, true) else BMID_PLAIN(128, false) }
#undef BMID_PLAIN
    SGLK_CHECK_LAUNCH("gemm_bf16_mid");
    return SGLK_OK;
}

// bf16 fused_experts below the 256-row kernel's range: tile table built with tile_m = 96 (kMidTileM); GATE_UP out = ic1
// [position][N] (p.N = N, p.K = K, n_half = N), DOWN out = ic2 [slot][K] (p.N = K, p.K = N)
int launch_moe_gemm_bf16_mid(int mode, const BmidParams& p, int max_mtiles, hipStream_t stream) {
    const int kblocks = p.K >> 7;
    if (p.K % 128 != 0 || p.N % 128 != 0 || kblocks < 2 || !p.tile_info || !p.num_tiles || !p.sorted_slot)
        SGLK_FAIL(SGLK_ERR_SHAPE, "moe_gemm_bf16_mid: N=%d K=%d not supported", p.N, p.K);
    const int64_t blocks = (int64_t)max_mtiles * (p.N >> 7);
    if (blocks == 0) return SGLK_OK;
    constexpr int TM = kMidTileM;
    const size_t lds = 2 * TM * 256 + 2 * TM * 4;
    const bool odd = (kblocks & 1) != 0;
    if (mode == MODE_GATE_UP) {
        // decode-size launches: four-wave workgroups up to half the CUs, activations three K blocks ahead up to two rounds of the
        // chip (policy and knobs of moe_gemm_fp8w_mid.hip)
        const int cus = device_cu_count();
        int nw = 8;
        if (!knobs().no_mid_narrow) {
            if (blocks * 2 <= cus) nw = 4;
            if (knobs().mid_nw == 4 || knobs().mid_nw == 8) nw = knobs().mid_nw;
        }
        const bool far = !knobs().no_mid_narrow && !odd && kblocks >= 4 && (knobs().mid_far >= 0 ? knobs().mid_far == 1 : blocks < 2 * (int64_t)cus);
        const int64_t nb = blocks * (8 / nw);
        if (nw == 4 && odd) hipLaunchKernelGGL((gbmid::gemm_bf16_mid_kernel<MODE_GATE_UP, TM, true, 4, 1>), dim3((unsigned)nb), dim3(256), lds, stream, p);
        else if (nw == 4 && far) hipLaunchKernelGGL((gbmid::gemm_bf16_mid_kernel<MODE_GATE_UP, TM, false, 4, 3>), dim3((unsigned)nb), dim3(256), lds, stream, p);
        else if (nw == 4) hipLaunchKernelGGL((gbmid::gemm_bf16_mid_kernel<MODE_GATE_UP, TM, false, 4, 1>), dim3((unsigned)nb), dim3(256), lds, stream, p);
        else if (far) hipLaunchKernelGGL((gbmid::gemm_bf16_mid_kernel<MODE_GATE_UP, TM, false, 8, 3>), dim3((unsigned)nb), dim3(512), lds, stream, p);
        else if (odd) hipLaunchKernelGGL((gbmid::gemm_bf16_mid_kernel<MODE_GATE_UP, TM, true>), dim3((unsigned)blocks), dim3(512), lds, stream, p);
        else hipLaunchKernelGGL((gbmid::gemm_bf16_mid_kernel<MODE_GATE_UP, TM, false>), dim3((unsigned)blocks), dim3(512), lds, stream, p);
    } else {
        if (odd) hipLaunchKernelGGL((gbmid::gemm_bf16_mid_kernel<MODE_DOWN, TM, true>), dim3((unsigned)blocks), dim3(512), lds, stream, p);
        else hipLaunchKernelGGL((gbmid::gemm_bf16_mid_kernel<MODE_DOWN, TM, false>), dim3((unsigned)blocks), dim3(512), lds, stream, p);
    }
    SGLK_CHECK_LAUNCH("moe_gemm_bf16_mid");
    return SGLK_OK;
}

}  // namespace sglk
