// Grouped (per-expert) W8A16 GEMMs of fused_experts: fp8-e4m3 weights x bf16 activations on bf16 MFMA.
//
// Replaces the two GEMM stages inside the reference's fused_experts_cpu(use_fp8_w8a16=True)
// (/root/reference/bench_moe.py:113-130); semantics = the oracle /root/reference/test_moe_fp8_ext.py:70-91:
//   MODE_GATE_UP : ic1[p, n]  = silu(x_p . W1[e][n]) * (x_p . W1[e][N+n])          (GEMM-1 + SiLU*mul)
//   MODE_DOWN    : ic2[s, k]  = topk_w[s] * (ic1[p] . W2[e][k]),  s = sorted_slot[p] (GEMM-2, scattered by slot)
// with W[r][c] = fp8(r,c) * scale[r / block_n][c / 128]  (block scale, fp32).
//
// Numerics: fp8 -> bf16 is exact, products are exact in fp32, one fp32 MFMA accumulation chain per 128-wide
// K block; the block scale multiplies that partial sum in fp32 (acc += s * partial).  No weight is ever rounded
// after scaling, which is tighter than "dequantise to bf16 then multiply".
//
// Tiling (v1): workgroup = 128 tokens x 128 weight rows, 4 waves as 2(n) x 2(m), each wave 64 x 64 =
// 4 x 4 tiles of mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand (so a lane's 4 accumulator
// registers are 4 consecutive output columns of ONE token -> 8-byte packed bf16 stores and gate/up meet in the
// same lane).  K is walked in 64-deep stages, two LDS buffers, both operands arrive by LDS-DMA
// (global_load_lds_dwordx4):
//   X stage  128 rows x 128 B, 16-byte chunks XOR-swizzled by (row & 7) on the SOURCE address (rows are gathered
//            through sorted_slot, so every lane carries its own row pointer anyway) -> conflict-free ds_read_b128
//   W stage  8 packed tiles x 1 KiB, already in lane order (pack.hip) -> one ds_read_b128 = both k-steps
#include "sglk_common.h"
#include "moe_internal.h"

namespace sglk {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

SGLK_DEV void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

// 8 fp8 (two dwords) -> 8 bf16, exact
SGLK_DEV bf16x8 fp8x8_to_bf16x8(unsigned lo, unsigned hi) {
    const bf16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false);
    const bf16x2 b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
    const bf16x2 c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false);
    const bf16x2 d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
    bf16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1];
    r[4] = c[0]; r[5] = c[1]; r[6] = d[0]; r[7] = d[1];
    return r;
}

constexpr int kStageX = kTileM * 128;                 // 16 KiB: 128 rows x 64 bf16
constexpr int kStageW = 8 * 1024;                     // 8 KiB : 8 packed 16x64 fp8 tiles
constexpr int kStage = kStageX + kStageW;

template <int MODE>
__global__ __launch_bounds__(256, 2) void moe_gemm_fp8w_kernel(const MoeGemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kStage];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;

    // the grid is sized for the worst case; only the first num_tiles*n_tiles blocks have work.  The XCD remap is
    // taken over THAT count, so every XCD gets an equal contiguous share of the real tiles.
    const int live = p.num_tiles[0] * p.n_tiles;
    if ((int)blockIdx.x >= live) return;
    const int L = xcd_remap(blockIdx.x, live);
    const int mtile = L / p.n_tiles;
    const int ntile = L - mtile * p.n_tiles;
    const int4 ti = p.tile_info[mtile];
    const int e = __builtin_amdgcn_readfirstlane(ti.x);
    const int pos0 = __builtin_amdgcn_readfirstlane(ti.y);
    const int rows = __builtin_amdgcn_readfirstlane(ti.z);

    const int ctiles = p.C >> 6;   // packed tiles (and stages) along the reduction dim

    // ---- per-lane LDS-DMA sources -------------------------------------------------------------------
    // X: this wave moves pieces q = 4*wave + i (8 rows x 128 B each); lane -> row 8q + lane/8, physical chunk lane%8
    const unsigned char* xsrc[4];
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
        const int logical_chunk = (lane & 7) ^ (r & 7);
        xsrc[i] = reinterpret_cast<const unsigned char*>(p.x + xrow) + logical_chunk * 16;
    }
    // W: this wave moves packed tiles 2*wave, 2*wave+1 of the workgroup's 8
    const unsigned char* wsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = wave * 2 + i;
        int row16;
        if (MODE == MODE_GATE_UP) {
            row16 = (piece < 4) ? ntile * 4 + piece : (p.n_half >> 4) + ntile * 4 + (piece - 4);
        } else {
            row16 = ntile * 8 + piece;
        }
        wsrc[i] = p.w + (int64_t)e * p.w_expert_stride + ((int64_t)row16 * ctiles) * 1024 + lane * 16;
    }

    auto issue_stage = [&](int kt, int buf) {
        unsigned char* sx = smem + buf * kStage;
        unsigned char* sw = sx + kStageX;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(xsrc[i] + (int64_t)kt * 128, sx + (wave * 4 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(wsrc[i] + (int64_t)kt * 1024, sw + (wave * 2 + i) * 1024);
    };

    // ---- per-lane LDS read offsets ---------------------------------------------------------------------
    int wt_piece[4], srow[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        int row16;
        if (MODE == MODE_GATE_UP) {
            wt_piece[nt] = (nt < 2) ? wn * 2 + nt : 4 + wn * 2 + (nt - 2);
            row16 = (nt < 2) ? ntile * 4 + wn * 2 + nt : (p.n_half >> 4) + ntile * 4 + wn * 2 + (nt - 2);
        } else {
            wt_piece[nt] = wn * 4 + nt;
            row16 = ntile * 8 + wn * 4 + nt;
        }
        srow[nt] = (row16 * 16) / p.block_n;
    }
    const float* scale_e = p.w_scale + (int64_t)e * p.scale_rows * p.scale_cols;

    int xoff[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int r = wm * 64 + mt * 16 + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xoff[mt][ks] = r * 128 + (((ks * 4 + (lane >> 4)) ^ (r & 7)) << 4);
    }

    f32x4 acc[4][4], tacc[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf, bool first) {
        const unsigned char* sx = smem + buf * kStage;
        const unsigned char* sw = sx + kStageX;
        bf16x8 wf[4][2];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const u32x4 raw = *reinterpret_cast<const u32x4*>(sw + wt_piece[nt] * 1024 + lane * 16);
            wf[nt][0] = fp8x8_to_bf16x8(raw[0], raw[1]);
            wf[nt][1] = fp8x8_to_bf16x8(raw[2], raw[3]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xf[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) xf[mt] = *reinterpret_cast<const bf16x8*>(sx + xoff[mt][ks]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    if (first && ks == 0)
                        tacc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][0], xf[mt], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else
                        tacc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], xf[mt], tacc[nt][mt], 0, 0, 0);
                }
        }
    };

    const int kblocks = p.C >> 7;
    issue_stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kb = 0; kb < kblocks; ++kb) {
        float s[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) s[nt] = scale_e[srow[nt] * p.scale_cols + kb];

        issue_stage(2 * kb + 1, 1);
        compute(0, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kb + 1 < kblocks) issue_stage(2 * kb + 2, 0);
        compute(1, false);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[nt][mt] += s[nt] * tacc[nt][mt];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------------------------------
    const int q4 = (lane >> 4) * 4;   // the lane's 4 accumulator registers = weight rows q4..q4+3 of the 16-row tile
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int r = wm * 64 + mt * 16 + (lane & 15);
        if (r >= rows) continue;
        if (MODE == MODE_GATE_UP) {
            uint16_t* orow = p.out + (int64_t)(pos0 + r) * p.out_stride + ntile * 64 + wn * 32 + q4;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f32x4 g = acc[nt][mt], u = acc[nt + 2][mt];
                uint2 v;
                v.x = pack_bf16x2(silu_f32(g[0]) * u[0], silu_f32(g[1]) * u[1]);
                v.y = pack_bf16x2(silu_f32(g[2]) * u[2], silu_f32(g[3]) * u[3]);
                *reinterpret_cast<uint2*>(orow + nt * 16) = v;
            }
        } else {
            const int slot = p.sorted_slot[pos0 + r];
            const float tw = p.topk_weights[slot];
            uint16_t* orow = p.out + (int64_t)slot * p.out_stride + ntile * 128 + wn * 64 + q4;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 a = acc[nt][mt];
                uint2 v;
                v.x = pack_bf16x2(a[0] * tw, a[1] * tw);
                v.y = pack_bf16x2(a[2] * tw, a[3] * tw);
                *reinterpret_cast<uint2*>(orow + nt * 16) = v;
            }
        }
    }
}

int launch_moe_gemm_fp8w(int mode, const MoeGemmParams& p, int max_mtiles, hipStream_t stream) {
    const int64_t blocks = (int64_t)max_mtiles * p.n_tiles;
    if (blocks == 0) return SGLK_OK;
    if (mode == MODE_GATE_UP)
        hipLaunchKernelGGL(moe_gemm_fp8w_kernel<MODE_GATE_UP>, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL(moe_gemm_fp8w_kernel<MODE_DOWN>, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    SGLK_CHECK_LAUNCH("moe_gemm_fp8w");
    return SGLK_OK;
}

}  // namespace sglk
