// Generic grouped / dense GEMM engine: any shape, any of the reference's weight types, packed or row-major weights.
//
// This is the kernel behind every GEMM-shaped operator that is NOT the tuned fp8 hot path:
//   fused_experts_cpu  bf16 / int8-w8a8 / fp8 with shapes the tuned tiles cannot take, is_vnni=False
//                      (/root/reference/test_moe.py:79-92, test_moe_int8.py:129, test_moe_offloading_cpu.py:123-137)
//   shared_expert_cpu  (/root/reference/test_moe_fp8.py:87-88, test_shared_experts.py:68,78)
//   fp8_scaled_mm_cpu, int8_scaled_mm_cpu, weight_packed_linear
//                      (/root/reference/test_gemm_fp8.py:54-62, test_gemm_int8.py:67, test_gemm.py:22-25)
//
// Both operands are converted EXACTLY to bf16 while they are staged into LDS (fp8 e4m3 and int8 values are exactly
// representable in bf16) and multiplied on mfma_f32_16x16x32_bf16 with fp32 accumulation, so every scale is applied
// in fp32 to fp32 sums: fp8 block scales to the per-K-block partial sum (acc += s * partial), int8 per-token x
// per-channel scales in the epilogue.  That is the arithmetic of the reference's oracles
// (/root/reference/test_gemm_int8.py:25-47: float matmul of the int8 values, then As * C * Bs).
//
// Tile: 64 tokens x 64 weight rows x 64 k, 256 threads; weights are the MFMA A operand (see moe_gemm_fp8w.hip).
// Register-staged (global -> VGPR -> convert -> LDS), single LDS buffer: built for coverage, not for the roofline —
// every access is bounds-guarded (rows, columns and the K tail are zero-filled).
#include "sglk_common.h"
#include "moe_internal.h"

namespace sglk {

namespace gg {

constexpr int kBM = kGenericTileM;   // 64 tokens
constexpr int kBN = 64;              // weight rows per tile
constexpr int kBK = 64;

// 8 consecutive reduction elements of one row -> 8 bf16 (16 bytes), exact.  `k` is a multiple of 8.
template <int TYPE>
SGLK_DEV uint4 load8_as_bf16(const unsigned char* row_base, int k, int C) {
    uint4 out = make_uint4(0, 0, 0, 0);
    if (k >= C) return out;
    if (TYPE == SGLK_W_BF16) {
        const unsigned short* p = reinterpret_cast<const unsigned short*>(row_base) + k;
        if (k + 8 <= C && (reinterpret_cast<uintptr_t>(p) & 15) == 0) return *reinterpret_cast<const uint4*>(p);
        unsigned short v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (k + j < C) ? p[j] : (unsigned short)0;
        out.x = v[0] | ((unsigned)v[1] << 16); out.y = v[2] | ((unsigned)v[3] << 16);
        out.z = v[4] | ((unsigned)v[5] << 16); out.w = v[6] | ((unsigned)v[7] << 16);
        return out;
    }
    const unsigned char* p = row_base + k;
    unsigned char b[8];
    if (k + 8 <= C && (reinterpret_cast<uintptr_t>(p) & 7) == 0) {
        const uint2 raw = *reinterpret_cast<const uint2*>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) { b[j] = (raw.x >> (8 * j)) & 0xff; b[4 + j] = (raw.y >> (8 * j)) & 0xff; }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = (k + j < C) ? p[j] : (unsigned char)0;
    }
    float f[8];
    if (TYPE == SGLK_W_FP8_E4M3) {
        const unsigned lo = b[0] | (b[1] << 8) | (b[2] << 16) | ((unsigned)b[3] << 24);
        const unsigned hi = b[4] | (b[5] << 8) | (b[6] << 16) | ((unsigned)b[7] << 24);
        const f32x2 a0 = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), a1 = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
        const f32x2 a2 = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), a3 = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
        f[0] = a0[0]; f[1] = a0[1]; f[2] = a1[0]; f[3] = a1[1]; f[4] = a2[0]; f[5] = a2[1]; f[6] = a3[0]; f[7] = a3[1];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (float)(signed char)b[j];
    }
    out.x = pack_bf16x2(f[0], f[1]); out.y = pack_bf16x2(f[2], f[3]);
    out.z = pack_bf16x2(f[4], f[5]); out.w = pack_bf16x2(f[6], f[7]);
    return out;
}

// MX-fp4 (/root/reference/test_mxfp4.py:14-127): 8 consecutive E2M1 values of one row (4 bytes, element 2i in the low
// nibble of byte i) times the row's E8M0 block scale 2^(s-127) of the 32-wide block -> 8 bf16.  Exact: an E2M1 value has
// two significant bits and the scale is a power of two (v_cvt_scalef32_pk_bf16_fp4 does both in one instruction).
// scales: [R][C/32] row-major, or (scale_packed) the reference's convert_scale_packed order [R/32][C/32][32].
SGLK_DEV uint4 load8_mxfp4(const unsigned char* row_base, const unsigned char* scales, int scale_packed, int row, int k, int C) {
    if (k >= C) return make_uint4(0, 0, 0, 0);
    const unsigned raw = *reinterpret_cast<const unsigned*>(row_base + (k >> 1));
    const int kb = k >> 5, nkb = C >> 5;
    const int64_t si = scale_packed ? ((int64_t)(row >> 5) * nkb + kb) * 32 + (row & 31) : (int64_t)row * nkb + kb;
    const unsigned sb = scales[si];
    const float sc = __uint_as_float(sb ? sb << 23 : 0x00400000u);   // 2^(sb-127); sb = 0 is the fp32 denormal 2^-127
    const bf16x2 a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(raw, sc, 0);
    const bf16x2 b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(raw, sc, 1);
    const bf16x2 c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(raw, sc, 2);
    const bf16x2 d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(raw, sc, 3);
    uint4 out;
    out.x = __builtin_bit_cast(unsigned, a); out.y = __builtin_bit_cast(unsigned, b);
    out.z = __builtin_bit_cast(unsigned, c); out.w = __builtin_bit_cast(unsigned, d);
    return out;
}

// the k-octet (row, k..k+7) of ONE packed matrix [R][C] (pack.hip orders) as 8 bf16; k % 8 == 0, k < C
template <int TYPE>
SGLK_DEV uint4 load_packed_octet(const unsigned char* mat, int row, int k, int C) {
    if (TYPE == SGLK_W_BF16) {
        // reference VNNI-2 order [R/32][C/2][32][2]: the octet is four (k, k+1) pairs, 32 pairs apart
        const unsigned* pairs = reinterpret_cast<const unsigned*>(mat) + ((int64_t)(row >> 5) * (C >> 1) + (k >> 1)) * 32 + (row & 31);
        return make_uint4(pairs[0], pairs[32], pairs[64], pairs[96]);
    }
    const int64_t tile = (int64_t)(row >> 4) * (C >> 6) + (k >> 6);
    const int kk = k & 63;
    int64_t off;
    if (TYPE == SGLK_W_FP8_E4M3) off = tile * 1024 + ((((kk & 31) >> 3) * 16 + (row & 15)) * 16) + (kk >> 5) * 8;
    else off = tile * 1024 + (((kk >> 4) * 16 + (row & 15)) * 16) + (kk & 8);
    return load8_as_bf16<TYPE>(mat + off, 0, 8);
}

template <int WTYPE, int XTYPE, int MODE>
__global__ __launch_bounds__(256) void gemm_generic_kernel(const GenericGemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kBM * 128];
    unsigned char* sx = smem;
    unsigned char* sw = smem + kBM * 128;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int n_mtiles = p.dense_rows > 0 ? (p.dense_rows + kBM - 1) / kBM : p.num_tiles[0];
    const int live = n_mtiles * p.n_tiles;
    const bool split = (MODE == GG_PLAIN) && p.ksplit > 1;
    const int ksr = split ? (int)(blockIdx.x % (unsigned)p.ksplit) : 0;            // reduction range of this workgroup
    const int bid = split ? (int)(blockIdx.x / (unsigned)p.ksplit) : (int)blockIdx.x;
    if (bid >= live) return;
    const int mtile = bid / p.n_tiles;
    const int ntile = bid - mtile * p.n_tiles;
    int4 ti;
    if (p.dense_rows > 0) {
        const int left = p.dense_rows - mtile * kBM;
        ti = make_int4(0, mtile * kBM, left < kBM ? left : kBM, 0);
    } else {
        ti = p.tile_info[mtile];
    }
    const int e = ti.x, pos0 = ti.y, rows = ti.z;

    const int C = p.C;
    constexpr int kOutCols = (MODE == GG_GATE_UP) ? kBN / 2 : kBN;    // output columns per tile
    const int col0 = ntile * kOutCols;

    // weight row of tile row i (0..63); -1 = out of range (zero-filled)
    auto weight_row = [&](int i) -> int {
        if (MODE == GG_GATE_UP) {
            const int c = col0 + (i & 31);
            if (c >= p.n_out) return -1;
            return (i < 32) ? c : p.n_half + c;
        }
        const int c = col0 + i;
        return c < p.n_out ? c : -1;
    };

    // ---- staging assignments: 512 octet-chunks per operand tile, 2 per thread -------------------------------------
    int x_row[2], x_ch[2], w_row[2];
    const unsigned char* x_base[2];
    const unsigned char* w_base[2];
    const unsigned char* wexp = reinterpret_cast<const unsigned char*>(p.w) + (int64_t)e * p.w_expert_stride;
    constexpr int WES = (WTYPE == SGLK_W_BF16) ? 2 : 1;
    constexpr int XES = (XTYPE == SGLK_W_BF16) ? 2 : 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i;
        x_row[i] = c >> 3;
        x_ch[i] = c & 7;
        const int r = x_row[i];
        x_base[i] = nullptr;
        if (r < rows) {
            int64_t xr;
            if (p.gather == GG_GATHER_TOKEN) xr = p.sorted_slot[pos0 + r] / p.topk;
            else xr = pos0 + r;
            x_base[i] = reinterpret_cast<const unsigned char*>(p.x) + xr * p.x_stride * XES;
        }
        w_row[i] = weight_row(r);
        if (WTYPE == SGLK_W_MXFP4) w_base[i] = w_row[i] >= 0 ? wexp + (int64_t)w_row[i] * (C >> 1) : nullptr;
        else w_base[i] = (w_row[i] >= 0 && !p.packed) ? wexp + (int64_t)w_row[i] * C * WES : nullptr;
    }

    // ---- fragment read offsets --------------------------------------------------------------------------------------
    // wave w: token tile w (16 tokens), all four 16-row weight tiles
    const int fr = lane & 15, fg = lane >> 4;
    int xoff[2], woff[4][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int xr = wave * 16 + fr;
        xoff[ks] = xr * 128 + (((ks * 4 + fg) ^ (xr & 7)) << 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int wr = nt * 16 + fr;
            woff[nt][ks] = wr * 128 + (((ks * 4 + fg) ^ (wr & 7)) << 4);
        }
    }

    f32x4 acc[4], tacc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; tacc[nt] = acc[nt]; }

    // fp8 block scale of the wave's weight tile nt for K block kb (tile rows share a block: block_n % 16 == 0)
    const float* scale_e = (WTYPE == SGLK_W_FP8_E4M3) ? p.w_scale + (int64_t)e * p.scale_rows * p.scale_cols : nullptr;
    int srow[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int wr = weight_row(nt * 16);
        srow[nt] = (WTYPE == SGLK_W_FP8_E4M3 && wr >= 0) ? wr / p.block_n : 0;
    }

    const int stages_all = (C + kBK - 1) / kBK;
    const int kt0 = split ? ksr * p.split_stages : 0;                               // split_stages is even (fp8 K blocks)
    const int stages = split ? (kt0 + p.split_stages < stages_all ? kt0 + p.split_stages : stages_all) : stages_all;
    // the operands of stage kt+1 are fetched into registers while stage kt is multiplied out of LDS
    uint4 xv[2], wv[2];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = kt * kBK + x_ch[i] * 8;
            xv[i] = x_base[i] ? load8_as_bf16<XTYPE>(x_base[i], k, C) : make_uint4(0, 0, 0, 0);
            if (w_row[i] < 0) {
                wv[i] = make_uint4(0, 0, 0, 0);
            } else if (WTYPE == SGLK_W_MXFP4) {   // p.w_scale = the E8M0 bytes, p.block_n = their layout flag
                wv[i] = load8_mxfp4(w_base[i], reinterpret_cast<const unsigned char*>(p.w_scale), p.block_n, w_row[i], k, C);
            } else if (p.packed) {
                // packed shapes have C % 64 == 0 (fp8/int8) or C % 8 == 0 (bf16): octets are complete
                wv[i] = (k < C) ? load_packed_octet<WTYPE>(wexp, w_row[i], k, C) : make_uint4(0, 0, 0, 0);
            } else {
                wv[i] = load8_as_bf16<WTYPE>(w_base[i], k, C);
            }
        }
    };
    if (kt0 < stages) fetch(kt0);
    for (int kt = kt0; kt < stages; ++kt) {
        __syncthreads();   // previous stage's fragment reads are done
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int off = x_row[i] * 128 + ((x_ch[i] ^ (x_row[i] & 7)) << 4);
            *reinterpret_cast<uint4*>(sx + off) = xv[i];
            *reinterpret_cast<uint4*>(sw + off) = wv[i];
        }
        __syncthreads();
        if (kt + 1 < stages) fetch(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(sx + xoff[ks]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(sw + woff[nt][ks]);
                if (WTYPE == SGLK_W_FP8_E4M3) tacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, tacc[nt], 0, 0, 0);
                else acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[nt], 0, 0, 0);
            }
        }
        if (WTYPE == SGLK_W_FP8_E4M3 && ((kt & 1) == 1 || kt == stages - 1)) {
            const int kb = kt >> 1;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                acc[nt] += scale_e[srow[nt] * p.scale_cols + kb] * tacc[nt];
                tacc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    const int r = wave * 16 + fr;          // token row of this lane inside the tile
    if (r >= rows) return;
    const int q4 = fg * 4;                 // the lane's 4 registers = weight rows q4..q4+3 of each 16-row tile
    int64_t xrow_idx;                       // row index into x (for the int8 activation scale)
    int slot = 0;
    if (p.gather == GG_GATHER_TOKEN || p.scatter) slot = p.sorted_slot[pos0 + r];
    xrow_idx = (p.gather == GG_GATHER_TOKEN) ? slot / p.topk : pos0 + r;
    const float xs = (XTYPE == SGLK_W_INT8) ? p.x_row_scale[xrow_idx] : 1.f;
    const int64_t orow = p.scatter ? slot : (int64_t)(pos0 + r);
    const float* wcs = (WTYPE == SGLK_W_INT8) ? p.w_scale + (int64_t)e * p.scale_rows : nullptr;   // per weight row

    auto store = [&](int col, float v) {
        if (p.out_type == SGLK_OUT_F32) reinterpret_cast<float*>(p.out)[orow * p.out_stride + col] = v;
        else if (p.out_type == SGLK_OUT_F16) reinterpret_cast<_Float16*>(p.out)[orow * p.out_stride + col] = (_Float16)v;
        else reinterpret_cast<unsigned short*>(p.out)[orow * p.out_stride + col] = f32_to_bf16_bits(v);
    };

    if (MODE == GG_GATE_UP) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = col0 + nt * 16 + q4 + j;
                if (c >= p.n_out) continue;
                float g = acc[nt][j], u = acc[nt + 2][j];
                if (WTYPE == SGLK_W_INT8) { g = xs * g * wcs[c]; u = xs * u * wcs[p.n_half + c]; }   // oracle order: As * C * Bs
                store(c, silu_f32(g) * u);
            }
    } else {
        const float tw = (MODE == GG_DOWN) ? p.topk_weights[slot] : 1.f;
        if (split) {   // fp32 partial of this reduction range; bias / addend / cast happen in the reduce launch
            float* prow = p.partial + ((int64_t)ksr * p.split_rows + orow) * p.n_out;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = col0 + nt * 16 + q4 + j;
                    if (c >= p.n_out) continue;
                    float v = acc[nt][j];
                    if (WTYPE == SGLK_W_INT8) v = xs * v * wcs[c];
                    prow[c] = v;
                }
            return;
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = col0 + nt * 16 + q4 + j;
                if (c >= p.n_out) continue;
                float v = acc[nt][j];
                if (WTYPE == SGLK_W_INT8) v = xs * v * wcs[c];   // oracle order: As * C * Bs (test_gemm_int8.py:42)
                if (p.bias) v += p.bias[c];
                if (p.addend) v += bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(p.addend)[orow * p.addend_stride + c]) * p.addend_scale;
                store(c, v * tw);
            }
    }
}

// out[r][c] = cast(sum over ranges (ascending) of partial[range][r][c] + bias[c] + addend[r][c] * scale)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GenericGemmParams p, int rows) {
    const int64_t total = (int64_t)rows * p.n_out;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / p.n_out;
        const int c = (int)(i - r * p.n_out);
        float v = 0.f;
        for (int k = 0; k < p.ksplit; ++k) v += p.partial[((int64_t)k * p.split_rows + r) * p.n_out + c];
        if (p.bias) v += p.bias[c];
        if (p.addend) v += bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(p.addend)[r * p.addend_stride + c]) * p.addend_scale;
        if (p.moe_ic2) {   // routed experts' combine, folded in: fp32 sum over the valid slots in slot order
            float sum = 0.f;
            for (int j = 0; j < p.moe_topk; ++j) {
                const int e = p.moe_ids[r * p.moe_topk + j];
                if (e >= 0 && e < p.moe_E) sum += bf16_bits_to_f32(p.moe_ic2[(r * p.moe_topk + j) * p.n_out + c]);
            }
            v += sum * p.addend_scale;
        }
        if (p.out_type == SGLK_OUT_F32) reinterpret_cast<float*>(p.out)[r * p.out_stride + c] = v;
        else if (p.out_type == SGLK_OUT_F16) reinterpret_cast<_Float16*>(p.out)[r * p.out_stride + c] = (_Float16)v;
        else reinterpret_cast<unsigned short*>(p.out)[r * p.out_stride + c] = f32_to_bf16_bits(v);
    }
}

// shared expert, small M: ic1[r][c] = bf16(silu(sum_k P[k][r][c]) * sum_k P[k][r][n + c]) from the fp32 partials of the
// gate_up GEMM (2n columns), ranges summed in order
__global__ __launch_bounds__(256) void splitk_reduce_silu_mul_kernel(const float* __restrict__ partial, int ksplit, int rows, int n,
                                                                     unsigned short* __restrict__ out, int64_t out_stride) {
    const int64_t total = (int64_t)rows * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n;
        const int c = (int)(i - r * n);
        float g = 0.f, u = 0.f;
        for (int k = 0; k < ksplit; ++k) {
            const float* row = partial + ((int64_t)k * rows + r) * (2 * n);
            g += row[c];
            u += row[n + c];
        }
        out[r * out_stride + c] = f32_to_bf16_bits(silu_f32(g) * u);
    }
}

}  // namespace gg

int launch_splitk_reduce_silu_mul(const float* partial, int ksplit, int rows, int n, uint16_t* out, int64_t out_stride,
                                  hipStream_t stream) {
    const int64_t total = (int64_t)rows * n;
    if (total == 0) return SGLK_OK;
    int64_t rb = ceil_div(total, 256);
    if (rb > 2048) rb = 2048;
    hipLaunchKernelGGL(gg::splitk_reduce_silu_mul_kernel, dim3((unsigned)rb), dim3(256), 0, stream, partial, ksplit, rows, n,
                       out, out_stride);
    SGLK_CHECK_LAUNCH("split-K reduce (SiLU*mul)");
    return SGLK_OK;
}

// Split-K plan: small-M dense GEMMs put only ceil(M/64) * ceil(N/64) workgroups on 256 CUs and each walks the whole
// reduction with one synchronous load per stage; cutting K lifts the number of workgroups to ~512.  Ranges are whole
// pairs of 64-deep stages (fp8 K blocks) and at least four stages long; the fp32 partials are bounded to 64 MiB.
// (Round 1 stopped at 256 rows; tall narrow outputs -- 2048 x 320 x 7168, the 576-wide kv_a projection of MLA at prefill sizes -- have
// as few tiles as a decode batch and took 0.1-0.2 ms with one workgroup walking 112 stages: any M now, bounded by the tile count.)
int generic_ksplit(int M, int N, int K) {
    if (M <= 0) return 1;
    const int64_t tiles = ceil_div(M, kGenericTileM) * ceil_div(N, 64);
    const int stages = (int)ceil_div(K, 64);
    if (tiles >= (M > 256 ? 600 : 256) || stages < 8) return 1;
    int ks = (int)((M > 256 ? 1024 : 512) / tiles);
    if (ks > stages / 4) ks = stages / 4;
    if (ks > 32) ks = 32;
    while (ks > 1 && (int64_t)ks * M * N * 4 > (64ll << 20)) --ks;
    return ks < 2 ? 1 : ks;
}

// m-tile table of a dense [M] x ... problem: one "expert", rows in natural order
__global__ void dense_tiles_kernel(int M, int tile_m, int4* tile_info, int* num_tiles, int* identity_slots) {
    const int n = (M + tile_m - 1) / tile_m;
    if (identity_slots)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) identity_slots[i] = i;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int rows = M - i * tile_m < tile_m ? M - i * tile_m : tile_m;
        tile_info[i] = make_int4(0, i * tile_m, rows, 0);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) num_tiles[0] = n;
}

// ... with ksplit K ranges per m-tile: entry (range, first row, rows), ranges fastest so that the workgroups of one m-tile
// (which share its activation rows) are neighbours
__global__ void dense_tiles_ksplit_kernel(int M, int tile_m, int ksplit, int4* tile_info, int* num_tiles, int* identity_slots) {
    const int n = (M + tile_m - 1) / tile_m;
    if (identity_slots)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) identity_slots[i] = i;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * ksplit; i += gridDim.x * blockDim.x) {
        const int mt = i / ksplit, ks = i - mt * ksplit;
        const int rows = M - mt * tile_m < tile_m ? M - mt * tile_m : tile_m;
        tile_info[i] = make_int4(ks, mt * tile_m, rows, 0);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) num_tiles[0] = n * ksplit;
}

int launch_dense_tiles_ksplit(int M, int tile_m, int ksplit, int4* tile_info, int* num_tiles, int* identity_slots, hipStream_t stream) {
    const int n = identity_slots ? M : ((M + tile_m - 1) / tile_m) * ksplit;
    int blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(dense_tiles_ksplit_kernel, dim3(blocks), dim3(256), 0, stream, M, tile_m, ksplit, tile_info, num_tiles, identity_slots);
    SGLK_CHECK_LAUNCH("dense_tiles");
    return SGLK_OK;
}

int launch_dense_tiles(int M, int tile_m, int4* tile_info, int* num_tiles, int* identity_slots, hipStream_t stream) {
    const int n = identity_slots ? M : (M + tile_m - 1) / tile_m;
    int blocks = (n + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(dense_tiles_kernel, dim3(blocks), dim3(256), 0, stream, M, tile_m, tile_info, num_tiles, identity_slots);
    SGLK_CHECK_LAUNCH("dense_tiles");
    return SGLK_OK;
}

int launch_gemm_generic(int mode, const GenericGemmParams& p, int max_mtiles, hipStream_t stream) {
    const bool split = mode == GG_PLAIN && p.ksplit > 1;
    if (split) SGLK_REQUIRE(p.partial && p.split_stages > 0 && p.split_stages % 2 == 0 && !p.scatter && p.gather == GG_GATHER_NONE,
                            SGLK_ERR_INVALID, "gemm_generic: bad split-K parameters");
    const int64_t blocks = (int64_t)max_mtiles * p.n_tiles * (split ? p.ksplit : 1);
    if (blocks == 0) return SGLK_OK;
    const dim3 grid((unsigned)blocks), block(256);
#define GG_LAUNCH(WT, XT, MD) hipLaunchKernelGGL((gg::gemm_generic_kernel<WT, XT, MD>), grid, block, 0, stream, p)
#define GG_MODES(WT, XT)                                  \
    do {                                                  \
        if (mode == GG_GATE_UP) GG_LAUNCH(WT, XT, GG_GATE_UP); \
        else if (mode == GG_DOWN) GG_LAUNCH(WT, XT, GG_DOWN);  \
        else GG_LAUNCH(WT, XT, GG_PLAIN);                 \
    } while (0)
    if (p.w_type == SGLK_W_BF16 && p.x_type == SGLK_W_BF16) GG_MODES(SGLK_W_BF16, SGLK_W_BF16);
    else if (p.w_type == SGLK_W_FP8_E4M3 && p.x_type == SGLK_W_BF16) GG_MODES(SGLK_W_FP8_E4M3, SGLK_W_BF16);
    else if (p.w_type == SGLK_W_INT8 && p.x_type == SGLK_W_INT8) GG_MODES(SGLK_W_INT8, SGLK_W_INT8);
    else if (p.w_type == SGLK_W_MXFP4 && p.x_type == SGLK_W_BF16 && mode == GG_PLAIN) GG_LAUNCH(SGLK_W_MXFP4, SGLK_W_BF16, GG_PLAIN);
    else SGLK_FAIL(SGLK_ERR_INVALID, "gemm_generic: unsupported operand types w=%d x=%d", p.w_type, p.x_type);
#undef GG_MODES
#undef GG_LAUNCH
    SGLK_CHECK_LAUNCH("gemm_generic");
    if (split) {
        return launch_splitk_reduce(p, stream);
    }
    return SGLK_OK;
}

// A caller that fuses the reduce into its own next kernel (qkv_proj.hip) asks for the split-K partials instead of the reduce
// launch: while a capture is set on this thread, the reduce is skipped and its description handed over.
static thread_local SplitkCapture* g_splitk_capture = nullptr;
void set_splitk_capture(SplitkCapture* c) { g_splitk_capture = c; }

int launch_splitk_reduce(const GenericGemmParams& p, hipStream_t stream) {
    const int64_t total = (int64_t)p.split_rows * p.n_out;
    if (total == 0) return SGLK_OK;
    if (g_splitk_capture && !p.bias && !p.addend && !p.moe_ic2 && p.out_type == SGLK_OUT_BF16) {
        g_splitk_capture->partial = p.partial;
        g_splitk_capture->ksplit = p.ksplit;
        g_splitk_capture->rows = p.split_rows;
        g_splitk_capture->n = p.n_out;
        return SGLK_OK;
    }
    int64_t rb = ceil_div(total, 256);
    if (rb > 2048) rb = 2048;
    hipLaunchKernelGGL(gg::splitk_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, stream, p, p.split_rows);
    SGLK_CHECK_LAUNCH("split-K reduce");
    return SGLK_OK;
}

}  // namespace sglk
