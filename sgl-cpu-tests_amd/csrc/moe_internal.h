// Internal declarations shared by the fused_experts stages (not part of the C-ABI).
#pragma once
#include "sglk_common.h"

namespace sglk {

constexpr int kTileM = 128;   // tokens (slot rows) per grouped-GEMM tile

enum { MODE_GATE_UP = 0, MODE_DOWN = 1 };

struct MoeGemmParams {
    const uint16_t* x;            // activations, bf16 bits
    int64_t x_stride;             // elements per row
    int64_t x_bytes;              // extent of x in bytes (buffer-descriptor range; < 4 GiB for the 256 kernel)
    const int* sorted_slot;       // [M*topk] slots grouped by expert (moe_align)
    int topk;
    const uint8_t* w;             // packed weights [E][R][C]
    int64_t w_expert_stride;      // bytes per expert
    const float* w_scale;         // [E][scale_rows][scale_cols]
    int scale_rows, scale_cols;
    int block_n;                  // scale block height in weight rows (block_k == 128)
    int C;                        // reduction length
    int n_half;                   // GATE_UP: N = row offset of the "up" half of w1
    const int4* tile_info;        // {expert, first position, rows, 0} per m-tile
    const int* num_tiles;
    int n_tiles;                  // workgroup tiles along the output columns
    uint16_t* out;                // GATE_UP: ic1 [position][N];  DOWN: ic2 [slot][K]
    int64_t out_stride;
    const float* topk_weights;    // DOWN only
};

int launch_moe_gemm_fp8w(int mode, const MoeGemmParams& p, int max_mtiles, hipStream_t stream);
// 256-token x 256-row tiles, 8 waves, 3-deep LDS-DMA ring (moe_gemm_fp8w_256.hip); tile table built with tile_m = 256
int launch_moe_gemm_fp8w_256(int mode, const MoeGemmParams& p, int max_mtiles, hipStream_t stream);

// out[m] = sum over valid slots j (ascending) of ic2[m*topk + j], fp32 sum, one bf16 rounding
int launch_moe_combine(const uint16_t* ic2, const int32_t* topk_ids, uint16_t* out, int64_t out_stride, int M,
                       int K, int E, int topk, hipStream_t stream);

}  // namespace sglk
