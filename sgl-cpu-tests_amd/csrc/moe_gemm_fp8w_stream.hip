// Small-M grouped W8A16 GEMM of fused_experts: weight-STREAMING kernel for experts that see at most a few dozen rows
// (decode and small prefill, M <~ 1500 at Qwen3-30B-A3B: every call has to read all touched experts' fp8 weights
// once, 604 MB when all 128 are hit, and does almost no math on them -> HBM-bound).
//
// Same math contract as moe_gemm_fp8w.hip (oracle /root/reference/test_moe_fp8_ext.py:22-25,70-91).
//
// * tile = 32 tokens (two 16-column MFMA B tiles) x 16 weight row-tiles per workgroup of 8 waves; every wave owns
//   two 16-row weight tiles (GATE_UP: the gate tile and the matching up tile, so SiLU*mul stays in-register);
// * the token tile's activations are gathered ONCE into LDS for the whole reduction length (32 x K bf16, <= 128 KiB,
//   16-byte chunks XOR-swizzled by row&15) and re-read from there by all waves;
// * weights never touch LDS: a packed tile piece (16 rows x 64 k = 1 KiB, pack.hip) is exactly one
//   global_load_dwordx4 per lane and lands in registers already in MFMA A-operand order.  Each wave keeps a ring of
//   kDepth pieces in flight (statically indexed registers), issued before the activation tile is even loaded, so
//   the stream never waits on a barrier: 8 waves x kDepth KiB per CU in flight;
// * fp8 -> bf16 exactly (v_cvt_scalef32_pk_bf16_fp8, scale 1.0), per-128-K-block partial accumulator scaled in fp32.
#include "knobs.h"
#include "moe_internal.h"
#include "moe_align_inline.h"

namespace sglk {
namespace gstream {

constexpr int kBM = kStreamTileM;   // 32 tokens
constexpr int kDepth = 8;           // weight pieces in flight per wave (8 x 16 B per lane = 32 VGPRs)

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

template <int MODE, bool NT>
__global__ __launch_bounds__(512, 2) void moe_gemm_fp8w_stream_kernel(const MoeGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xlds[];   // [32 tokens][C] bf16, swizzled

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // tile table: moe_align's, or (p.inline_ids, at most 32 slots) derived here by every wave from the ids themselves
    __shared__ int slot_tab[kInlineAlignSlots];   // inline form: slot of the tile's row r
    const bool inl = p.inline_ids != nullptr;
    InlineAlign ia{};
    if (inl) ia = inline_align(p.inline_ids, p.inline_slots, p.inline_experts, lane);
    const int live = (inl ? ia.ntiles : p.num_tiles[0]) * p.n_tiles;
    if ((int)blockIdx.x >= live) return;
    const int L = xcd_remap(blockIdx.x, live);
    const int mtile = L / p.n_tiles;
    const int ntile = L - mtile * p.n_tiles;
    int e, pos0, rows;
    if (inl) {
        const InlineTile it = inline_tile(ia, mtile, lane, wave == 0, slot_tab);   // exactly one expert: mtile < ntiles
        e = it.e; pos0 = it.pos0; rows = it.rows;
        __syncthreads();
    } else {
        const int4 ti = p.tile_info[mtile];
        e = __builtin_amdgcn_readfirstlane(ti.x);
        pos0 = __builtin_amdgcn_readfirstlane(ti.y);
        rows = __builtin_amdgcn_readfirstlane(ti.z);
    }
    auto slot_of = [&](int r) __attribute__((always_inline)) { return inl ? slot_tab[r] : p.sorted_slot[pos0 + r]; };

    const int C = p.C;
    const int ctiles = C >> 6;           // 64-wide pieces along the reduction dim
    const int row_bytes = C * 2;

    // ---- the wave's two weight row-tiles and their piece stream (tile0,kc0),(tile1,kc0),(tile0,kc1),... ------------
    int row16[2];
    if (MODE == MODE_GATE_UP) {          // workgroup = 128 ic1 columns: wave w -> columns ntile*128 + 16w .. +15
        row16[0] = ntile * 8 + wave;
        row16[1] = (p.n_half >> 4) + ntile * 8 + wave;
    } else {                             // workgroup = 256 output columns: wave w -> columns ntile*256 + 32w .. +31
        row16[0] = ntile * 16 + wave * 2;
        row16[1] = row16[0] + 1;
    }
    const unsigned char* wexp = p.w + (int64_t)e * p.w_expert_stride;
    const unsigned char* wp0 = wexp + ((int64_t)row16[0] * ctiles) * 1024 + lane * 16;
    const unsigned char* wp1 = wexp + ((int64_t)row16[1] * ctiles) * 1024 + lane * 16;
    const int npieces = 2 * ctiles;
    auto piece_ptr = [&](int i) { return ((i & 1) ? wp1 : wp0) + (int64_t)(i >> 1) * 1024; };

    u32x4 ring[kDepth];   // npieces >= kDepth: C >= 256
#pragma unroll
    for (int i = 0; i < kDepth; ++i) ring[i] = ld_stream16<NT>(piece_ptr(i));

    // ---- activations of the token tile -> LDS, whole reduction length -----------------------------------------------
    {
        const int chunks = C >> 3;                       // 16-byte chunks per row
        for (int c = tid; c < kBM * chunks; c += 512) {
            const int r = c / chunks, ch = c - r * chunks;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < rows) {
                int64_t xrow;
                if (MODE == MODE_GATE_UP) xrow = (int64_t)(slot_of(r) / p.topk) * p.x_stride;
                else xrow = (int64_t)(pos0 + r) * p.x_stride;
                v = *reinterpret_cast<const uint4*>(p.x + xrow + ch * 8);
            }
            *reinterpret_cast<uint4*>(xlds + r * row_bytes + ((ch ^ (r & 15)) << 4)) = v;
        }
    }
    __syncthreads();

    const float* scale_e = p.w_scale + (int64_t)e * p.scale_rows * p.scale_cols;
    const int srow0 = (row16[0] * 16) / p.block_n, srow1 = (row16[1] * 16) / p.block_n;

    const int r = lane & 15, g = lane >> 4;
    f32x4 acc[2][2], tacc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f}; tacc[a][b] = acc[a][b]; }

    // one piece = weight tile (i&1), 64-wide k chunk (i>>1): 2 k-steps x 2 token tiles
    auto consume = [&](const u32x4& raw, int i) {
        const int kc = i >> 1;
        const bf16x8 w0 = cvt8(raw[0], raw[1]), w1 = cvt8(raw[2], raw[3]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int xr = mt * 16 + r;
            const unsigned char* base = xlds + xr * row_bytes;
            const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(base + (((kc * 8 + g) ^ (xr & 15)) << 4));
            const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(base + (((kc * 8 + 4 + g) ^ (xr & 15)) << 4));
            f32x4 t = tacc[i & 1][mt];
            t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x0, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x1, t, 0, 0, 0);
            tacc[i & 1][mt] = t;
        }
        if (kc & 1) {   // second half of a 128-wide K block: fold the partial sum in with the block scale
            const float s = scale_e[((i & 1) ? srow1 : srow0) * p.scale_cols + (kc >> 1)];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[i & 1][mt] += s * tacc[i & 1][mt];
                tacc[i & 1][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    // ring walk.  npieces is a multiple of kDepth (C % 256 == 0), so the steady state has NO branch inside the
    // unrolled body: with control flow between a load and its use hipcc falls back to `s_waitcnt vmcnt(0)` per piece
    // and the ring degenerates to one piece in flight.  The last kDepth pieces are consumed without refills.
    int base = 0;
    for (; base + kDepth < npieces; base += kDepth) {
#pragma unroll
        for (int j = 0; j < kDepth; ++j) {
            const u32x4 raw = ring[j];
            ring[j] = ld_stream16<NT>(piece_ptr(base + j + kDepth));
            consume(raw, base + j);
        }
    }
#pragma unroll
    for (int j = 0; j < kDepth; ++j) consume(ring[j], base + j);

    // ---- epilogue ---------------------------------------------------------------------------------------------------------
    const int q4 = g * 4;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int tr = mt * 16 + r;
        if (tr >= rows) continue;
        if (MODE == MODE_GATE_UP) {
            const f32x4 gt = acc[0][mt], up = acc[1][mt];
            uint2 v;
            v.x = pack_bf16x2(silu_f32(gt[0]) * up[0], silu_f32(gt[1]) * up[1]);
            v.y = pack_bf16x2(silu_f32(gt[2]) * up[2], silu_f32(gt[3]) * up[3]);
            *reinterpret_cast<uint2*>(p.out + (int64_t)(pos0 + tr) * p.out_stride + ntile * 128 + wave * 16 + q4) = v;
        } else {
            const int slot = slot_of(tr);
            const float tw = p.topk_weights[slot];
            uint16_t* orow = p.out + (int64_t)slot * p.out_stride + ntile * 256 + wave * 32 + q4;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const f32x4 c = acc[a][mt] * tw;
                uint2 v;
                v.x = pack_bf16x2(c[0], c[1]);
                v.y = pack_bf16x2(c[2], c[3]);
                *reinterpret_cast<uint2*>(orow + a * 16) = v;
            }
        }
    }
}

}  // namespace gstream

int launch_moe_gemm_fp8w_stream(int mode, const MoeGemmParams& p, int max_mtiles, hipStream_t stream) {
    const int64_t blocks = (int64_t)max_mtiles * p.n_tiles;
    if (blocks == 0) return SGLK_OK;
    const size_t lds = (size_t)kStreamTileM * p.C * 2;
    if (lds > 150 * 1024) SGLK_FAIL(SGLK_ERR_SHAPE, "moe_gemm_fp8w_stream: reduction length %d too long for the LDS tile", p.C);
#define STREAM_LAUNCH(MD, NTV)                                                                                     \
    {                                                                                                              \
        SGLK_ENSURE_DYN_LDS((gstream::moe_gemm_fp8w_stream_kernel<MD, NTV>), 150 * 1024, "moe_gemm_fp8w_stream");  \
        hipLaunchKernelGGL((gstream::moe_gemm_fp8w_stream_kernel<MD, NTV>), dim3((unsigned)blocks), dim3(512), lds, stream, p); \
    }
    if (mode == MODE_GATE_UP) {
        if (p.w_nt) STREAM_LAUNCH(MODE_GATE_UP, true) else STREAM_LAUNCH(MODE_GATE_UP, false)
    } else {
        if (p.w_nt) STREAM_LAUNCH(MODE_DOWN, true) else STREAM_LAUNCH(MODE_DOWN, false)
    }
#undef STREAM_LAUNCH
    SGLK_CHECK_LAUNCH("moe_gemm_fp8w_stream");
    return SGLK_OK;
}

}  // namespace sglk
