// convert_weight_packed: re-tile row-major weights into MFMA-operand order (DESIGN.md §Packed weight layout).
//
// One packed tile is 1 KiB = 64 lanes x 16 B, i.e. exactly what one wave moves with a single
// global_load_lds_dwordx4 / global_load_dwordx4, and what one ds_read_b128 per lane turns into operands.
// Lane l of a tile (r = l & 15 -> weight row inside the 16-row tile, g = l >> 4 -> k group):
//   fp8  tile 16 rows x 64 cols: bytes 0..7  = W[r][64*ct +      8g .. +7]   (k-step 0 of mfma_16x16x32)
//                                bytes 8..15 = W[r][64*ct + 32 + 8g .. +7]   (k-step 1)
//   int8 tile 16 rows x 64 cols: bytes 0..15 = W[r][64*ct + 16g .. +15]     (one mfma_i32_16x16x64_i8 operand)
//   bf16 tile 16 rows x 32 cols: 8 elems     = W[r][32*ct + 8g .. +7]       (one mfma_16x16x32_bf16 operand)
// Tiles are stored [row tile][col tile], so a weight row-tile streams contiguously along the reduction dim.
#include "sglk_common.h"

namespace sglk {

template <int WTYPE, bool UNPACK>
__global__ __launch_bounds__(256) void pack_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                   int64_t rows, int64_t cols, int64_t total_chunks) {
    constexpr int ES = (WTYPE == SGLK_W_BF16) ? 2 : 1;        // element size
    constexpr int TC = (WTYPE == SGLK_W_BF16) ? 32 : 64;      // tile columns
    const int64_t ctiles = cols / TC;
    const int64_t tiles_per_mat = (rows / 16) * ctiles;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total_chunks;
         c += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(c & 63);
        const int64_t tile = c >> 6;
        const int64_t b = tile / tiles_per_mat;
        const int64_t t = tile - b * tiles_per_mat;
        const int64_t rt = t / ctiles, ct = t - rt * ctiles;
        const int r = lane & 15, g = lane >> 4;
        // row-major side: matrix b of the batch; packed side: chunk c sits at byte 16*c of the whole batch
        const uint8_t* mat = src + b * rows * cols * ES;   // pack: read from here
        uint8_t* omat = dst + b * rows * cols * ES;        // unpack: write to here
        const int64_t row = rt * 16 + r;
        const uint8_t* packed_in = src + c * 16;
        uint8_t* packed_ptr = dst + c * 16;
        if (WTYPE == SGLK_W_FP8_E4M3) {
            const int64_t off0 = (row * cols + ct * 64 + 8 * g) * ES;
            const int64_t off1 = off0 + 32;
            if (!UNPACK) {
                uint2 lo = *reinterpret_cast<const uint2*>(mat + off0);
                uint2 hi = *reinterpret_cast<const uint2*>(mat + off1);
                *reinterpret_cast<uint4*>(packed_ptr) = make_uint4(lo.x, lo.y, hi.x, hi.y);
            } else {
                uint4 v = *reinterpret_cast<const uint4*>(packed_in);
                *reinterpret_cast<uint2*>(omat + off0) = make_uint2(v.x, v.y);
                *reinterpret_cast<uint2*>(omat + off1) = make_uint2(v.z, v.w);
            }
        } else {
            const int64_t off = (row * cols + ct * TC) * ES + (int64_t)g * 16;
            if (!UNPACK) {
                *reinterpret_cast<uint4*>(packed_ptr) = *reinterpret_cast<const uint4*>(mat + off);
            } else {
                *reinterpret_cast<uint4*>(omat + off) = *reinterpret_cast<const uint4*>(packed_in);
            }
        }
    }
}

static int pack_common(const void* src, void* dst, int64_t batch, int64_t rows, int64_t cols, int wtype,
                       void* stream, bool unpack) {
    SGLK_REQUIRE(src && dst, SGLK_ERR_INVALID, "pack_weight: null pointer");
    SGLK_REQUIRE(src != dst, SGLK_ERR_INVALID, "pack_weight: in-place packing is not supported");
    SGLK_REQUIRE(batch >= 0 && rows >= 0 && cols >= 0, SGLK_ERR_INVALID, "pack_weight: negative size");
    SGLK_REQUIRE(wtype == SGLK_W_BF16 || wtype == SGLK_W_FP8_E4M3 || wtype == SGLK_W_INT8, SGLK_ERR_INVALID,
                 "pack_weight: unknown weight type %d", wtype);
    const int tc = (wtype == SGLK_W_BF16) ? 32 : 64;
    SGLK_REQUIRE(rows % 16 == 0 && cols % tc == 0, SGLK_ERR_SHAPE,
                 "pack_weight: rows (%lld) must be a multiple of 16 and cols (%lld) of %d", (long long)rows,
                 (long long)cols, tc);
    const int es = (wtype == SGLK_W_BF16) ? 2 : 1;
    const int64_t chunks = batch * rows * cols * es / 16;
    if (chunks == 0) return SGLK_OK;
    const int threads = 256;
    int64_t blocks = ceil_div(chunks, threads);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t s = (hipStream_t)stream;
    const uint8_t* sp = (const uint8_t*)src;
    uint8_t* dp = (uint8_t*)dst;
#define LAUNCH(WT)                                                                                        \
    if (unpack)                                                                                           \
        hipLaunchKernelGGL((pack_kernel<WT, true>), dim3((unsigned)blocks), dim3(threads), 0, s, sp, dp, rows, cols, chunks); \
    else                                                                                                  \
        hipLaunchKernelGGL((pack_kernel<WT, false>), dim3((unsigned)blocks), dim3(threads), 0, s, sp, dp, rows, cols, chunks)
    if (wtype == SGLK_W_BF16) { LAUNCH(SGLK_W_BF16); }
    else if (wtype == SGLK_W_FP8_E4M3) { LAUNCH(SGLK_W_FP8_E4M3); }
    else { LAUNCH(SGLK_W_INT8); }
#undef LAUNCH
    SGLK_CHECK_LAUNCH("pack_weight");
    return SGLK_OK;
}

}  // namespace sglk

extern "C" int sglk_pack_weight(const void* src, void* dst, int64_t batch, int64_t rows, int64_t cols, int wtype,
                                void* stream) {
    return sglk::pack_common(src, dst, batch, rows, cols, wtype, stream, false);
}

extern "C" int sglk_unpack_weight(const void* src, void* dst, int64_t batch, int64_t rows, int64_t cols, int wtype,
                                  void* stream) {
    return sglk::pack_common(src, dst, batch, rows, cols, wtype, stream, true);
}
