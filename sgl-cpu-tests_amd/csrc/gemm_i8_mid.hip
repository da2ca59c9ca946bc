// Small / mid-batch int8 W8A8 fused_experts (below the 256-row kernel's range): the weight-streaming scheme of
// moe_gemm_fp8w_mid.hip on the int8 matrix cores.  Oracle: /root/reference/test_moe_int8.py:16-94 (per-token dynamic
// activation quantisation, i8 x i8 -> i32, (As * acc) * Bs).
//
// * tile = up to 128 rows x 16 (GATE_UP) or 8 (DOWN) weight row-tiles per workgroup of 8 waves; a wave owns the gate and the
//   matching up tile (GATE_UP) or one tile (DOWN) for all the rows; MT = 4 or 8 column tiles of 16 rows by the tile's rows;
// * weights never touch LDS: a packed piece (16 rows x 64 k = 1 KiB, pack.hip) is one global_load_dwordx4 per lane and IS the
//   A operand of mfma_i32_16x16x64_i8; two 128-wide K blocks (four pieces per tile) are in flight per wave;
// * the quantised activations go global -> LDS by DMA one K block at a time (rows x 128 B, 16-byte chunks XOR-swizzled by
//   row & 7 on the source side), double buffered, one counted s_waitcnt (a builtin, so that the compiler's wait-count pass
//   sees it) + one barrier per block;
// * integer accumulation is exact; the epilogue applies (x_scale * acc) * w_scale in that order without contraction, like
//   gemm_i8_256.hip, then SiLU*mul -> fp32 ic1 (GATE_UP) or the routing weight -> bf16 rows by slot (DOWN).
#include "knobs.h"
#include "sglk_common.h"
#include "moe_internal.h"

#pragma clang fp contract(off)

namespace sglk {

typedef const __attribute__((address_space(1))) void* gptr_im_t;
typedef __attribute__((address_space(3))) void* lptr_im_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace gimid {

constexpr int kTM = kI8MidTileM;      // 128 rows at most
constexpr int kXBuf = kTM * 128;      // one K block of the row tile: 16 KiB
constexpr int kLds = 2 * kXBuf;

struct Ctx {
    int pos0, rows, ntile, kblocks, e, kb0, ksr;
    const unsigned char* wp[2];   // lane's 16 bytes inside the first piece of the wave's weight row-tiles
    int row0[2];                  // first weight row of each tile (for the per-row scales)
};

// NW = waves per workgroup (8; 4 for GATE_UP launches that reach at most half the CUs), XD = K blocks the activations travel ahead
// (1; 3 with four LDS buffers of the 64 rows a short tile has): as in moe_gemm_fp8w_mid.hip, where the two are measured
template <int MODE, int MT, bool ODD, int NW = 8, int XD = 1, bool NT = false>
SGLK_DEV void run(const I8GemmParams& p, unsigned char* lds, const Ctx& c) {
    static_assert(NW == 8 || MODE == MODE_GATE_UP, "narrow workgroups exist for GATE_UP only");
    static_assert(XD == 1 || (XD == 3 && MT == 4 && !ODD && MODE == MODE_GATE_UP), "far prefetch: short GATE_UP tiles, even block counts");
    constexpr int XB = XD + 1;
    constexpr int kXSz = XD == 1 ? kXBuf : MT * 16 * 128;
    constexpr int TPW = MODE == MODE_GATE_UP ? 2 : 1;
    constexpr int PB = 2 * TPW;          // weight pieces per K block per wave
    u32x4 ring[2 * PB];                  // slot = block*PB + tile*2 + k half
#pragma unroll
    for (int i = 0; i < 2 * PB; ++i)
        ring[i] = ld_stream16<NT>(c.wp[(i % PB) >> 1] + (int64_t)(2 * (i / PB) + (i & 1)) * 1024);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    constexpr int XV = MT * 2 / NW;      // 1-KiB DMA pieces (8 rows x 128 B) of a K block per wave: MT*16 rows / 8 / NW

    const int8_t* xsrc[XV];
#pragma unroll
    for (int j = 0; j < XV; ++j) {
        const int row = (wave * XV + j) * 8 + (lane >> 3), ch = (lane & 7) ^ (row & 7);
        const int rr = row < c.rows ? row : c.rows - 1;   // padding rows re-read the last row; their outputs are dropped
        int64_t xrow;
        if (MODE == MODE_GATE_UP) xrow = (int64_t)(p.sorted_slot[c.pos0 + rr] / p.topk) * p.x_stride;
        else xrow = (int64_t)(c.pos0 + rr) * p.x_stride;
        xsrc[j] = p.x + xrow + ch * 16 + (int64_t)c.kb0 * 128;
    }
    auto x_dma = [&](int kb) __attribute__((always_inline)) {
        unsigned char* dst = lds + (kb % XB) * kXSz + wave * XV * 1024;
#pragma unroll
        for (int j = 0; j < XV; ++j)
            __builtin_amdgcn_global_load_lds((gptr_im_t)(xsrc[j] + kb * 128), (lptr_im_t)(dst + j * 1024), 16, 0, 0);
    };

    i32x4 acc[TPW][MT];
#pragma unroll
    for (int a = 0; a < TPW; ++a)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[a][mt] = (i32x4){0, 0, 0, 0};

    // has_x: block kb+XD exists (requested here); prefetch_x: block kb+1 exists (sync at the end of this block)
    auto block = [&](int kb, int half, bool refill, bool prefetch_x, bool has_x) __attribute__((always_inline)) {
        if (has_x) x_dma(kb + XD);
        __builtin_amdgcn_sched_barrier(0);
        i32x4 w[TPW][2];
#pragma unroll
        for (int a = 0; a < TPW; ++a) {
            w[a][0] = __builtin_bit_cast(i32x4, ring[half * PB + 2 * a]);
            w[a][1] = __builtin_bit_cast(i32x4, ring[half * PB + 2 * a + 1]);
        }
        if (refill) {
#pragma unroll
            for (int j = 0; j < PB; ++j)
                ring[half * PB + j] = ld_stream16<NT>(c.wp[j >> 1] + (int64_t)(2 * (kb + 2) + (j & 1)) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* xb = lds + (kb % XB) * kXSz;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int xr = mt * 16 + r;
            const unsigned char* base = xb + xr * 128;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const i32x4 x = *reinterpret_cast<const i32x4*>(base + (((s * 4 + g) ^ (xr & 7)) << 4));
#pragma unroll
                for (int a = 0; a < TPW; ++a) acc[a][mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w[a][s], x, acc[a][mt], 0, 0, 0);
            }
        }
        if (prefetch_x) {
            if (XD == 3 && has_x && refill) __builtin_amdgcn_s_waitcnt(0x0F70 | (XV + PB));   // this block's own requests stay in flight
            else if (refill) __builtin_amdgcn_s_waitcnt(0x0F70 | PB);   // vmcnt(PB): all but the refills -> the DMA has landed
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();
        }
    };

    x_dma(0);
    if (XD == 3) {
        x_dma(1);
        x_dma(2);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
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
    } else if (!ODD) {
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

    // ---- epilogue: lane holds weight rows 4g..4g+3 of each tile for row r of every column tile ----------------------------
    const int q4 = g * 4;
    const float* wsc = p.w_scale + (int64_t)c.e * p.scale_rows;
    const float4 ws0 = *reinterpret_cast<const float4*>(wsc + c.row0[0] + q4);
    const float4 ws1 = TPW == 2 ? *reinterpret_cast<const float4*>(wsc + c.row0[1] + q4) : ws0;
    const int col = c.ntile * (NW * 16) + wave * 16 + q4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int tr = mt * 16 + r;
        if (tr >= c.rows) continue;
        if (MODE == MODE_PLAIN) {
            if (p.partial_i32) {  // exact int32 partial of this K range; scales, bias and the rounding belong to the reduce
                int32_t* dst = p.partial_i32 + ((int64_t)c.ksr * p.M + c.pos0 + tr) * p.N + col;
                *reinterpret_cast<i32x4*>(dst) = acc[0][mt];
            } else {
                const float xs = p.x_scale[c.pos0 + tr];
                float o4[4];
                o4[0] = xs * (float)acc[0][mt][0] * ws0.x;
                o4[1] = xs * (float)acc[0][mt][1] * ws0.y;
                o4[2] = xs * (float)acc[0][mt][2] * ws0.z;
                o4[3] = xs * (float)acc[0][mt][3] * ws0.w;
                if (p.bias) { o4[0] += p.bias[col]; o4[1] += p.bias[col + 1]; o4[2] += p.bias[col + 2]; o4[3] += p.bias[col + 3]; }
                const int64_t o = (int64_t)(c.pos0 + tr) * p.out_stride + col;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (p.out_type == SGLK_OUT_F32) reinterpret_cast<float*>(p.out)[o + i] = o4[i];
                    else if (p.out_type == SGLK_OUT_F16) reinterpret_cast<_Float16*>(p.out)[o + i] = (_Float16)o4[i];
                    else p.out[o + i] = f32_to_bf16_bits(o4[i]);
                }
            }
        } else if (MODE == MODE_GATE_UP) {
            const int slot = p.sorted_slot[c.pos0 + tr];
            const float xs = p.x_scale[slot / p.topk];
            float4 v;
            v.x = silu_f32(xs * (float)acc[0][mt][0] * ws0.x) * (xs * (float)acc[TPW - 1][mt][0] * ws1.x);
            v.y = silu_f32(xs * (float)acc[0][mt][1] * ws0.y) * (xs * (float)acc[TPW - 1][mt][1] * ws1.y);
            v.z = silu_f32(xs * (float)acc[0][mt][2] * ws0.z) * (xs * (float)acc[TPW - 1][mt][2] * ws1.z);
            v.w = silu_f32(xs * (float)acc[0][mt][3] * ws0.w) * (xs * (float)acc[TPW - 1][mt][3] * ws1.w);
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (int64_t)(c.pos0 + tr) * p.out_stride + col) = v;
        } else {
            const int slot = p.sorted_slot[c.pos0 + tr];
            const float xs = p.x_scale[c.pos0 + tr];
            const float tw = p.topk_weights[slot];
            float o4[4];
            o4[0] = xs * (float)acc[0][mt][0] * ws0.x * tw;
            o4[1] = xs * (float)acc[0][mt][1] * ws0.y * tw;
            o4[2] = xs * (float)acc[0][mt][2] * ws0.z * tw;
            o4[3] = xs * (float)acc[0][mt][3] * ws0.w * tw;
            uint2 v;
            v.x = pack_bf16x2(o4[0], o4[1]);
            v.y = pack_bf16x2(o4[2], o4[3]);
            *reinterpret_cast<uint2*>(p.out + (int64_t)slot * p.out_stride + col) = v;
        }
    }
}

template <int MODE, bool ODD, int NW = 8, int XD = 1, bool NT = false>
__global__ __launch_bounds__(NW * 64, 2) void gemm_i8_mid_kernel(const I8GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nsplit = (MODE == MODE_PLAIN && p.ksplit > 1) ? p.ksplit : 1;
    // PLAIN (dense): the row tiles are rows [128 i, 128 i + 128) of M rows, one "expert"
    const int mtiles = MODE == MODE_PLAIN ? (p.M + kTM - 1) / kTM : p.num_tiles[0];
    const int live = mtiles * p.n_tiles * nsplit;
    if ((int)blockIdx.x >= live) return;
    const int Ls = xcd_remap(blockIdx.x, live);
    const int L = Ls / nsplit;
    const int mtile = L / p.n_tiles;
    int4 ti;
    if (MODE == MODE_PLAIN) ti = make_int4(0, mtile * kTM, p.M - mtile * kTM < kTM ? p.M - mtile * kTM : kTM, 0);
    else ti = p.tile_info[mtile];
    Ctx c;
    c.e = __builtin_amdgcn_readfirstlane(ti.x);
    c.ntile = L - mtile * p.n_tiles;
    c.pos0 = __builtin_amdgcn_readfirstlane(ti.y);
    c.rows = __builtin_amdgcn_readfirstlane(ti.z);
    c.ksr = Ls - L * nsplit;
    c.kblocks = nsplit > 1 ? p.split_kblocks : p.K >> 7;
    c.kb0 = c.ksr * c.kblocks;
    const int ctiles = p.K >> 6;
    // both modes: 128 output columns per workgroup, wave w -> columns ntile*128 + 16w .. +15 (GATE_UP: gate + matching up tile)
    c.row0[0] = c.ntile * (NW * 16) + wave * 16;
    c.row0[1] = (MODE == MODE_GATE_UP ? p.n_half : 0) + c.row0[0];
    const unsigned char* wexp = p.w + (int64_t)c.e * p.w_bytes;
    c.wp[0] = wexp + ((int64_t)(c.row0[0] >> 4) * ctiles + 2 * c.kb0) * 1024 + lane * 16;
    c.wp[1] = wexp + ((int64_t)(c.row0[1] >> 4) * ctiles + 2 * c.kb0) * 1024 + lane * 16;
    if (c.rows <= 64) run<MODE, 4, ODD, NW, XD, NT>(p, lds, c);
    else run<MODE, 8, ODD, NW, 1, NT>(p, lds, c);
}

// out[r][c] = cast((x_scale[r] * (float)(sum over ranges of the int32 partials)) * w_scale[c] + bias[c]): the integer sum is
// exact, the float operations are the oracle's, in its order (/root/reference/test_gemm_int8.py:25-47)
__global__ __launch_bounds__(256) void i8_splitk_reduce_kernel(const I8GemmParams p) {
    const int64_t total = (int64_t)p.M * p.N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / p.N;
        const int cidx = (int)(i - r * p.N);
        int32_t acc = 0;
        for (int k = 0; k < p.ksplit; ++k) acc += p.partial_i32[((int64_t)k * p.M + r) * p.N + cidx];
        float v = p.x_scale[r] * (float)acc * p.w_scale[cidx];
        if (p.bias) v += p.bias[cidx];
        if (p.out_type == SGLK_OUT_F32) reinterpret_cast<float*>(p.out)[r * p.out_stride + cidx] = v;
        else if (p.out_type == SGLK_OUT_F16) reinterpret_cast<_Float16*>(p.out)[r * p.out_stride + cidx] = (_Float16)v;
        else p.out[r * p.out_stride + cidx] = f32_to_bf16_bits(v);
    }
}

// shared expert, gate_up: ic1[r][c] = silu((xs[r] * (float)G) * ws[c]) * ((xs[r] * (float)U) * ws[n + c]) in fp32, G / U = the
// exact int32 sums of the gate / up column over the K ranges (/root/reference/test_shared_experts.py:11-53 through the int8
// linear of test_gemm_int8.py:25-47)
__global__ __launch_bounds__(256) void i8_splitk_reduce_silu_mul_kernel(const int32_t* __restrict__ partial, int ksplit, int rows,
                                                                        int n, const float* __restrict__ xs,
                                                                        const float* __restrict__ ws, float* __restrict__ out) {
    const int64_t total = (int64_t)rows * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n;
        const int c = (int)(i - r * n);
        int32_t g = 0, u = 0;
        for (int k = 0; k < ksplit; ++k) {
            const int32_t* row = partial + ((int64_t)k * rows + r) * (2 * n);
            g += row[c];
            u += row[n + c];
        }
        const float gv = xs[r] * (float)g * ws[c], uv = xs[r] * (float)u * ws[n + c];
        out[r * n + c] = silu_f32(gv) * uv;
    }
}

// shared expert, down: out[r][c] = bf16((xs[r] * (float)S) * ws[c] + addend[r][c] * scale)
__global__ __launch_bounds__(256) void i8_splitk_reduce_addend_kernel(const int32_t* __restrict__ partial, int ksplit, int rows, int n,
                                                                      const float* __restrict__ xs, const float* __restrict__ ws,
                                                                      const unsigned short* __restrict__ addend, int64_t addend_stride,
                                                                      float addend_scale, unsigned short* __restrict__ out,
                                                                      int64_t out_stride) {
    const int64_t total = (int64_t)rows * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n;
        const int c = (int)(i - r * n);
        int32_t acc = 0;
        for (int k = 0; k < ksplit; ++k) acc += partial[((int64_t)k * rows + r) * n + c];
        float v = xs[r] * (float)acc * ws[c];
        if (addend) v += bf16_bits_to_f32(addend[r * addend_stride + c]) * addend_scale;
        out[r * out_stride + c] = f32_to_bf16_bits(v);
    }
}

}  // namespace gimid

int launch_i8_reduce_silu_mul(const int32_t* partial, int ksplit, int rows, int n, const float* xs, const float* ws, float* out,
                              hipStream_t stream) {
    const int64_t total = (int64_t)rows * n;
    if (total == 0) return SGLK_OK;
    int64_t rb = ceil_div(total, 256);
    if (rb > 2048) rb = 2048;
    hipLaunchKernelGGL(gimid::i8_splitk_reduce_silu_mul_kernel, dim3((unsigned)rb), dim3(256), 0, stream, partial, ksplit, rows, n, xs, ws, out);
    SGLK_CHECK_LAUNCH("int8 split-K reduce (SiLU*mul)");
    return SGLK_OK;
}

int launch_i8_reduce_addend(const int32_t* partial, int ksplit, int rows, int n, const float* xs, const float* ws,
                            const uint16_t* addend, int64_t addend_stride, float addend_scale, uint16_t* out, int64_t out_stride,
                            hipStream_t stream) {
    const int64_t total = (int64_t)rows * n;
    if (total == 0) return SGLK_OK;
    int64_t rb = ceil_div(total, 256);
    if (rb > 2048) rb = 2048;
    hipLaunchKernelGGL(gimid::i8_splitk_reduce_addend_kernel, dim3((unsigned)rb), dim3(256), 0, stream, partial, ksplit, rows, n, xs, ws,
                       addend, addend_stride, addend_scale, out, out_stride);
    SGLK_CHECK_LAUNCH("int8 split-K reduce (addend)");
    return SGLK_OK;
}

// decode-size dense W8A8: one row tile (M <= 128), 128 output columns per workgroup, equal even K ranges until ~512 workgroups
int i8_mid_ksplit(int M, int N, int K) {
    if (M <= 0 || M > gimid::kTM || N % 128 != 0 || K % 256 != 0) return 0;
    const int kblocks = K >> 7;
    const int64_t tiles = N / 128;
    if (knobs().mid_dense_model > 0) return splitk_by_rounds(kblocks, tiles, M, N, device_cu_count() * (knobs().mid_dense_model == 1 ? 2 : 1));
    const int per_min = kblocks >= 4 ? 4 : 2;
    int best = 0;
    for (int per = kblocks; per >= per_min; per -= 2) {
        if (kblocks % per != 0) continue;
        const int ks = kblocks / per;
        if (ks > 32 || (int64_t)ks * M * N * 4 > (64ll << 20)) continue;
        best = ks;
        if (tiles * ks >= 512) break;
    }
    return best;
}

// Dense GEMMs above 128 rows: rows [128 i, 128 i + 128) are the kernel's PLAIN row tiles; the same range policy over all tiles.
// (The 256-row tile kernel has ceil(M / 256) x N / 256 workgroups and takes one tile's time -- K / 64 stages -- however few they are:
// 160 x 4096 x 4096 ran 55 us there against 20 us here at 128 rows.)
int i8_mid_dense_ksplit(int M, int N, int K) {
    if (M <= gimid::kTM) return i8_mid_ksplit(M, N, K);
    if (M >= kMidDenseMaxM || N % 128 != 0 || K % 256 != 0) return 0;
    const int kblocks = K >> 7;
    const int64_t tiles = (int64_t)ceil_div(M, gimid::kTM) * (N / 128);
    if (knobs().mid_dense_model > 0) return splitk_by_rounds(kblocks, tiles, M, N, device_cu_count() * (knobs().mid_dense_model == 1 ? 2 : 1));
    const int per_min = kblocks >= 4 ? 4 : 2;
    int best = 0;
    for (int per = kblocks; per >= per_min; per -= 2) {
        if (kblocks % per != 0) continue;
        const int ks = kblocks / per;
        if (ks > 32 || (int64_t)ks * M * N * 4 > (64ll << 20)) continue;
        best = ks;
        if (tiles * ks >= 512) break;
    }
    return best;
}

int launch_gemm_i8_mid_plain(const I8GemmParams& p, hipStream_t stream) {
    const int nsplit = p.ksplit > 1 ? p.ksplit : 1;
    const int kblocks = nsplit > 1 ? p.split_kblocks : p.K >> 7;
    if (p.K % 128 != 0 || p.N % 128 != 0 || p.M >= kMidDenseMaxM || kblocks < 2 || kblocks % 2 != 0 || p.x_stride % 16 != 0 ||
        (nsplit > 1 && ((p.K >> 7) != nsplit * kblocks || !p.partial_i32)) || (!p.out && !p.partial_i32))
        SGLK_FAIL(SGLK_ERR_SHAPE, "gemm_i8_mid(plain): M=%d N=%d K=%d with %d ranges not supported", p.M, p.N, p.K, nsplit);
    if (p.M == 0) return SGLK_OK;
    I8GemmParams q = p;
    q.n_tiles = p.N >> 7;
    q.scale_rows = p.N;
    const int64_t blocks = (int64_t)ceil_div(p.M, gimid::kTM) * q.n_tiles * nsplit;
    hipLaunchKernelGGL((gimid::gemm_i8_mid_kernel<MODE_PLAIN, false>), dim3((unsigned)blocks), dim3(512), gimid::kLds, stream, q);
    SGLK_CHECK_LAUNCH("gemm_i8_mid(plain)");
    if (nsplit > 1 && p.out) {   // p.out == nullptr: the caller reduces the partials itself (shared expert)
        int64_t rb = ceil_div((int64_t)p.M * p.N, 256);
        if (rb > 2048) rb = 2048;
        hipLaunchKernelGGL(gimid::i8_splitk_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, stream, q);
        SGLK_CHECK_LAUNCH("gemm_i8_mid(reduce)");
    }
    return SGLK_OK;
}

// the ordered reduce of exact int32 split-K partials [ksplit][M][N] with the oracle's float operations (also behind the 256-row
// int8 kernel's dense split-K)
int launch_i8_splitk_reduce(const I8GemmParams& q, hipStream_t stream) {
    int64_t rb = ceil_div((int64_t)q.M * q.N, 256);
    if (rb > 2048) rb = 2048;
    if (rb <= 0) return SGLK_OK;
    hipLaunchKernelGGL(gimid::i8_splitk_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, stream, q);
    SGLK_CHECK_LAUNCH("int8 split-K reduce");
    return SGLK_OK;
}

// tile table built with tile_m = kI8MidTileM; GATE_UP: n_tiles = N / 128, out = fp32 ic1 [position][N]; DOWN: n_tiles = R / 128,
// out = bf16 ic2 [slot][R]
int launch_gemm_i8_mid(int mode, const I8GemmParams& p, int max_mtiles, hipStream_t stream) {
    const int kblocks = p.K >> 7;
    if (p.K % 128 != 0 || kblocks < 2 || !p.tile_info || !p.num_tiles || !p.sorted_slot || p.x_stride % 16 != 0)
        SGLK_FAIL(SGLK_ERR_SHAPE, "gemm_i8_mid: K=%d not supported", p.K);
    const int64_t blocks = (int64_t)max_mtiles * p.n_tiles;
    if (blocks == 0) return SGLK_OK;
    const bool odd = (kblocks & 1) != 0;
    const size_t lds = gimid::kLds;
#define I8MID(MD, OD)                                                                                                            \
    {                                                                                                                            \
        if (p.w_nt) hipLaunchKernelGGL((gimid::gemm_i8_mid_kernel<MD, OD, 8, 1, true>), dim3((unsigned)blocks), dim3(512), lds, stream, p);  \
        else hipLaunchKernelGGL((gimid::gemm_i8_mid_kernel<MD, OD, 8, 1, false>), dim3((unsigned)blocks), dim3(512), lds, stream, p);        \
    }
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
        I8GemmParams q = p;
        q.n_tiles = p.n_tiles * (8 / nw);
        const int64_t nb = (int64_t)max_mtiles * q.n_tiles;
#define I8MIDN(OD, NWV, XDV) hipLaunchKernelGGL((gimid::gemm_i8_mid_kernel<MODE_GATE_UP, OD, NWV, XDV>), dim3((unsigned)nb), dim3(NWV * 64), lds, stream, q)
        if (nw == 4 && odd) I8MIDN(true, 4, 1);
        else if (nw == 4 && far) I8MIDN(false, 4, 3);
        else if (nw == 4) I8MIDN(false, 4, 1);
        else if (far) I8MIDN(false, 8, 3);
        else if (odd) I8MID(MODE_GATE_UP, true)
        else I8MID(MODE_GATE_UP, false)
#undef I8MIDN
    } else if (odd) I8MID(MODE_DOWN, true)
    else I8MID(MODE_DOWN, false)
#undef I8MID
    SGLK_CHECK_LAUNCH("gemm_i8_mid");
    return SGLK_OK;
}

}  // namespace sglk
