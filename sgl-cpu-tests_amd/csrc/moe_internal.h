// Internal declarations shared by the fused_experts stages (not part of the C-ABI).
#pragma once
#include "sglk_common.h"

namespace sglk {

constexpr int kTileM = 128;        // tokens (slot rows) per tile of the 128x128 kernel
constexpr int kStreamTileM = 32;   // tokens per tile of the weight-streaming small-M kernel
constexpr int kI8MidTileM = 128;   // rows per tile (at most) of the int8 weight-streaming kernel (gemm_i8_mid.hip)
constexpr int kMidDenseMaxM = 2048;   // dense / shared-expert forms of the weight-streaming kernels take fewer rows than this (the dispatch decides below it)
constexpr int kMidTileM = 96;      // tokens per tile (at most) of the weight-streaming mid-M kernel

enum { MODE_GATE_UP = 0, MODE_DOWN = 1, MODE_PLAIN = 2 };   // PLAIN: out[pos] = x.W^T (+bias) (+addend*scale), dense

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
    const float* bias;            // PLAIN: [output columns] f32 or null
    const uint16_t* addend;       // PLAIN: bf16 [rows][out columns] added as addend * addend_scale, or null
    int64_t addend_stride;
    float addend_scale;
    // PLAIN on the mid kernel (small-M dense): split-K -- workgroup = (tile, K range of split_kblocks 128-wide blocks), fp32
    // partial [range][row][out_cols] reduced by launch_splitk_reduce; ksplit <= 1: whole reduction, bf16 out (+ bias)
    int ksplit, split_kblocks, split_rows, out_cols;
    float* partial;
    int w_nt;               // decode-size kernels: read the weights with the non-temporal policy (sglk_common.h, ld_stream16)
    // PLAIN on the 256-row kernel with split-K (dense GEMMs whose tiles would not fill the chip): the tile table's "expert" is
    // the K range; C = the range's length, c_full = the whole reduction length (row stride of the packed weight / scale table),
    // w_expert_stride = bytes of one range inside a row tile, w_bytes_total = extent of the weight for the buffer descriptor
    int c_full;
    int64_t w_bytes_total;
    int* tickets;                 // persistent 256-tile kernel: 8 zeroed counters (one per XCD) or null = static tile split
    // weight-streaming kernel, at most 32 slots: no moe_align launch -- every workgroup sorts the ids itself (moe_align_inline.h)
    const int* inline_ids;        // topk_ids [inline_slots] or null = tile_info / num_tiles / sorted_slot tables
    int inline_slots, inline_experts;
    unsigned long long* dbg;      // developer builds only (SGLK_DEV_ABLATE): per-workgroup {shader clocks, 100 MHz ticks}
};

int launch_moe_gemm_fp8w(int mode, const MoeGemmParams& p, int max_mtiles, hipStream_t stream);
// 256-token x 256-row tiles, 8 waves, 3-deep LDS-DMA ring on mfma_f32_32x32x16_bf16, main loop issued MFMA by MFMA with
// the next k-step's feed in the shadows (moe_gemm_fp8w_256i.hip); tile table built with tile_m = 256; needs
// block_n % 32 == 0 and a reduction of at least two 128-wide K blocks
int launch_moe_gemm_fp8w_256i(int mode, const MoeGemmParams& p, int max_mtiles, hipStream_t stream);
bool moe_gemm_fp8w_256i_is_persistent(int C, int64_t tiles);   // what that launch does for `tiles` workgroup tiles

// ---- the fp8 / int8 matrix-core kernels on 128-token tiles (moe_gemm_fp8w_s128.hip): W8A16 two-term split, opt-in a8 mode, int8 ----
struct A8GemmParams {
    const uint8_t* x;             // quantised activations: e4m3, packed-tile k order inside every 64 group
    int64_t x_stride;             // bytes per row
    int64_t x_bytes;              // extent (< 4 GiB, buffer descriptor)
    const uint8_t* xs;            // E8M0 scale bytes [row][C / 128], row stride xs_stride (multiple of 4)
    int xs_stride;
    const int* sorted_slot;
    int topk;
    const uint8_t* w;             // packed fp8 weights [E][R][C]
    int64_t w_expert_stride;
    const float* w_scale;         // [E][scale_rows][scale_cols]
    int scale_rows, scale_cols, block_n;
    int C;                        // reduction length
    int n_half;                   // GATE_UP: N
    const int4* tile_info;        // tile_m = 128
    const int* num_tiles;
    int n_tiles;                  // GATE_UP: N / 128; DOWN: K / 256
    void* out;                    // GATE_UP: ic1 e4m3 [position][N] (same k order); DOWN: ic2 bf16 [slot][K]
    int64_t out_stride;           // GATE_UP: bytes per row; DOWN: elements per row
    uint8_t* out_s;               // GATE_UP: ic1 scale bytes [position][out_s_stride]
    int out_s_stride;
    const float* topk_weights;    // DOWN
    int max_mtiles;               // s128 kernel: entries of tile_info that may be read (the launch's m-tile bound)
    int prio;                     // s128 kernel: wave priority experiment (knobs().s128_prio; 0 = none)
    // s128 kernel, int8 W8A8 (terms = 0): x / ic1 rows are int8 in natural k order, w = pack.hip's int8 tiles, w_scale = per weight
    // row [E][scale_rows]; one f32 factor per x row instead of xs; GATE_UP writes int8 ic1 + out_scale_f32[position] and needs
    // row_amax [M * topk] (zeroed) and arrivals [max_mtiles] (zeroed) for the per-token maximum across the m-tile's workgroups
    const float* bias;            // s128 kernel, MODE_PLAIN (dense rows, no tile table: m-tile i = rows [128 i, 128 i + 128) of dense_rows): [output columns] f32 or null
    int dense_rows;
    const float* x_scale_f32;
    float* out_scale_f32;
    unsigned* row_amax;
    int* arrivals;
    float quant_floor;
    unsigned long long* dbg;      // developer builds only (SGLK_DEV_ABLATE): per-workgroup 100 MHz time stamps
};
// W8A16 with EXACT bf16 activations as two e4m3 terms (hi + lo) on the block-scaled fp8 matrix cores (fp8_split.h):
// x / ic1 rows are [hi 64 | lo 64] per 64-wide k group (x_stride / out_stride = 2 * C bytes), one E8M0 byte per 128 block;
// 128-token tiles, four waves, two workgroups per CU, weights streamed global -> VGPR
// (moe_gemm_fp8w_s128.hip); tile table built with tile_m = 128; GATE_UP n_tiles = N / 128, DOWN n_tiles = K / 256
// terms = 2: x / ic1 rows are the two-term split (above); terms = 1: the a8 mode's quantised rows (fp8_split.h: quant_row_block128);
// terms = 0: the int8 W8A8 operator (int8 rows, per-row f32 factors; see A8GemmParams)
int launch_moe_gemm_fp8w_s128(int mode, const A8GemmParams& p, int max_mtiles, hipStream_t stream, int terms = 2);
bool moe_gemm_fp8w_s128_ok(int N, int K, int block_n);
int launch_split_fp8_block128(const uint16_t* x, int64_t x_stride, uint8_t* q, int64_t q_stride, uint8_t* s, int64_t s_stride,
                              int64_t rows, int cols, hipStream_t stream);
// hidden bf16 [rows][cols] -> e4m3 (packed-tile k order) + one E8M0 byte per 128-wide block
int launch_quant_fp8_block128(const uint16_t* x, int64_t x_stride, uint8_t* q, int64_t q_stride, uint8_t* s, int64_t s_stride,
                              int64_t rows, int cols, hipStream_t stream);

// ---- dense W8A8 GEMM on the int8 matrix cores (gemm_i8_256.hip) ----------------------------------------------------------
struct I8GemmParams {
    const int8_t* x;          // [rows][K] int8, row stride x_stride bytes (multiple of 16)
    int64_t x_stride;
    int64_t x_bytes;          // extent of x (< 4 GiB, buffer descriptor)
    const float* x_scale;     // per x row
    const uint8_t* w;         // packed int8 [E][R][K] (pack.hip tile order)
    int64_t w_bytes;          // bytes per expert = R * K (< 4 GiB)
    const float* w_scale;     // [E][scale_rows] per weight row
    int scale_rows;           // R
    const float* bias;        // PLAIN: [R] or null
    uint16_t* out;            // PLAIN: bf16 [M][R]; GATE_UP: fp32 ic1 [position][N]; DOWN: bf16 ic2 [slot][R]
    int64_t out_stride;       // elements of the output type
    int M, K;                 // dense: rows of x; K = reduction length
    int n_tiles;              // PLAIN / DOWN: R / 256; GATE_UP: N / 128
    // grouped (fused_experts): m-tile table with tile_m = 256; null = dense
    const int4* tile_info;
    const int* num_tiles;
    const int* sorted_slot;
    int topk;
    int n_half;               // GATE_UP: N = row offset of the up half of w1
    const float* topk_weights;   // DOWN
    // PLAIN on the weight-streaming kernel (decode-size dense W8A8, gemm_i8_mid.hip): N output columns, K ranges of
    // split_kblocks 128-wide blocks with exact int32 partials [range][M][N] (ksplit <= 1: whole reduction, direct output)
    int N, ksplit, split_kblocks;
    int w_nt;               // MoE launches of the mid kernel: non-temporal weight reads (ld_stream16)
    int32_t* partial_i32;
    int out_type;             // PLAIN: SGLK_OUT_* of `out`
    // PLAIN on the 256-row kernel: out += addend[row][col] * addend_scale in fp32 before the bf16 rounding (shared expert), or null
    const uint16_t* addend;
    int64_t addend_stride;    // elements (multiple of 4)
    float addend_scale;
};
int launch_gemm_i8_256(int mode, const I8GemmParams& p, int max_mtiles, hipStream_t stream);
int launch_i8_splitk_reduce(const I8GemmParams& q, hipStream_t stream);   // [ksplit][M][N] int32 partials -> out (scales, bias)

// ---- bf16 GEMM with VNNI-2 packed weights on the bf16 matrix cores (gemm_bf16_256.hip) ------------------------------------
struct Bf16GemmParams {
    const uint16_t* x;        // bf16 [rows][K], row stride x_stride BYTES (multiple of 16)
    int64_t x_stride;
    int64_t x_bytes;          // extent of x (< 4 GiB, buffer descriptor)
    const uint8_t* w;         // packed bf16 [E][R/32][K/2][32][2]
    int64_t w_bytes;          // bytes per expert = R * K * 2 (< 4 GiB)
    const float* bias;        // PLAIN: [R] f32 or null
    uint16_t* out;            // bf16: PLAIN [M][R]; GATE_UP ic1 [position][N]; DOWN ic2 [slot][R]
    int64_t out_stride;       // elements
    int M, K;                 // dense: rows of x; K = reduction length (elements)
    int n_tiles;              // PLAIN / DOWN: R / 256; GATE_UP: N / 128
    const int4* tile_info;    // grouped: m-tile table with tile_m = 256; null = dense
    const int* num_tiles;
    const int* sorted_slot;
    int topk;
    int n_half;               // GATE_UP: N
    const float* topk_weights;   // DOWN
    // dense PLAIN with split-K (tiles that would not fill the chip): K = the length of one range, k_full = the whole reduction
    // (row-block stride of the packed weight), workgroup = (m-tile, column tile, range); fp32 partial [range][M][out_cols]
    int ksplit, k_full, out_cols;
    float* partial;
    // PLAIN without split-K: out += addend[row][col] * addend_scale in fp32 before the bf16 rounding (shared expert), or null
    const uint16_t* addend;
    int64_t addend_stride;    // elements (multiple of 4)
    float addend_scale;
};
int launch_gemm_bf16_256(int mode, const Bf16GemmParams& p, int max_mtiles, hipStream_t stream);

// ---- generic engine (gemm_generic.hip) ------------------------------------------------------------------------------
constexpr int kGenericTileM = 64;
enum { GG_GATE_UP = 0, GG_DOWN = 1, GG_PLAIN = 2 };
enum { GG_GATHER_NONE = 0, GG_GATHER_TOKEN = 1 };   // x row = position | sorted_slot[position] / topk

struct GenericGemmParams {
    const void* x;                // tokens: bf16 or int8 (x_type = SGLK_W_BF16 / SGLK_W_INT8)
    int x_type;
    int64_t x_stride;             // elements
    const float* x_row_scale;     // int8 activations: per x row
    const int* sorted_slot;
    int topk;
    int gather;
    const int4* tile_info;        // 64-row m-tiles
    const int* num_tiles;
    int dense_rows;               // > 0: one "expert", m-tile i = rows [64 i, 64 i + 64) of dense_rows -- no tile table is read
                                  // (saves the table-building launch in front of every decode-size dense GEMM)
    int n_tiles;                  // tiles along output columns
    const void* w;                // [E][R][C]
    int w_type;
    int packed;
    int64_t w_expert_stride;      // bytes
    int C;                        // reduction length
    const float* w_scale;         // fp8: [E][scale_rows][scale_cols] block scales; int8: [E][scale_rows] per weight row
    int scale_rows, scale_cols, block_n;
    int n_half;                   // GATE_UP: row offset of the "up" half
    int n_out;                    // output columns
    void* out;
    int out_type;                 // SGLK_OUT_*
    int64_t out_stride;
    int scatter;                  // 1: output row = slot (MoE down projection)
    const float* topk_weights;    // GG_DOWN
    const float* bias;            // [n_out] f32 or null
    const void* addend;           // bf16 [rows][n_out] added as addend * addend_scale (shared expert) or null
    int64_t addend_stride;
    float addend_scale;
    // split-K (GG_PLAIN, dense rows only): the reduction is cut into `ksplit` ranges of `split_stages` 64-deep stages, every
    // (tile, range) workgroup writes an fp32 partial [range][row][n_out] and a second launch sums the ranges in order and
    // applies bias / addend / the output cast.  ksplit <= 1: off.
    int ksplit, split_stages, split_rows;
    float* partial;
    // split-K reduce only (moe block): instead of `addend`, add addend_scale * sum over the valid routing slots j (ascending)
    // of moe_ic2[(row * moe_topk + j)][n_out] -- the routed experts' top-k combine folded into the shared expert's last launch
    const uint16_t* moe_ic2;
    const int32_t* moe_ids;
    int moe_topk, moe_E;
};

// int8 W8A8 fused_experts at small / mid batch sizes, weight-streaming (gemm_i8_mid.hip); grouped modes only
int launch_gemm_i8_mid(int mode, const I8GemmParams& p, int max_mtiles, hipStream_t stream);
// decode-size dense W8A8 (M <= 128): 0 = shape not taken, else the number of K ranges; the launch includes the exact reduce
int i8_mid_ksplit(int M, int N, int K);
int i8_mid_dense_ksplit(int M, int N, int K);   // dense form: several 128-row tiles (M < 1024); equals i8_mid_ksplit up to 128 rows
int launch_gemm_i8_mid_plain(const I8GemmParams& p, hipStream_t stream);   // p.out == nullptr: partials only (no reduce)
int launch_i8_reduce_silu_mul(const int32_t* partial, int ksplit, int rows, int n, const float* xs, const float* ws, float* out,
                              hipStream_t stream);
int launch_i8_reduce_addend(const int32_t* partial, int ksplit, int rows, int n, const float* xs, const float* ws,
                            const uint16_t* addend, int64_t addend_stride, float addend_scale, uint16_t* out, int64_t out_stride,
                            hipStream_t stream);

// decode-size dense bf16 GEMM on packed (VNNI-2) weights, weight-streaming (gemm_bf16_mid.hip)
struct BmidParams {
    const uint16_t* x;        // bf16 [M][K], row stride x_stride elements (multiple of 8), 16-byte aligned
    int64_t x_stride;
    const uint8_t* w;         // packed bf16 [N/32][K/2][32][2]
    const float* bias;        // [N] f32 or null (ksplit <= 1 only; with split-K the reduce adds it)
    uint16_t* out;            // bf16 [M][N]
    int64_t out_stride;
    int M, N, K;
    int ksplit, split_kblocks;   // K ranges of split_kblocks 128-wide blocks; <= 1: whole reduction
    float* partial;           // [ksplit][M][N] fp32
    // grouped modes (bf16 fused_experts at small / mid batch sizes): N = weight rows of the launch's output (GATE_UP: the
    // expert width, weights [E][2N][K]; DOWN: the hidden size), tile table built with tile_m = 96
    int64_t w_expert_stride;  // bytes per expert
    const int4* tile_info;
    const int* num_tiles;
    const int* sorted_slot;
    int topk;
    int n_half;               // GATE_UP: row offset of the "up" half
    const float* topk_weights;   // DOWN
};
int launch_moe_gemm_bf16_mid(int mode, const BmidParams& p, int max_mtiles, hipStream_t stream);
int bf16_mid_ksplit(int M, int N, int K);   // 0 = shape not taken, else the number of K ranges
int launch_gemm_bf16_mid(const BmidParams& p, hipStream_t stream);

// split-K plan of a small-M dense GEMM (same answer for the workspace size and for the launch): 1 = no split
int generic_ksplit(int M, int N, int K);

int launch_gemm_generic(int mode, const GenericGemmParams& p, int max_mtiles, hipStream_t stream);
// the ordered reduce of split-K partials alone (fields used: partial, ksplit, split_rows, n_out, out, out_type, out_stride,
// bias, addend*)
int launch_dense_tiles_ksplit(int M, int tile_m, int ksplit, int4* tile_info, int* num_tiles, int* identity_slots, hipStream_t stream);
int launch_splitk_reduce(const GenericGemmParams& p, hipStream_t stream);
// partial != nullptr after a dense GEMM call: the call left its fp32 split-K partials [ksplit][rows][n] un-reduced (see
// set_splitk_capture); else the GEMM wrote its bf16 output itself
struct SplitkCapture {
    const float* partial;
    int ksplit;
    int64_t rows;
    int n;
};
void set_splitk_capture(SplitkCapture* c);   // thread-local; nullptr switches it off
// ic1 = bf16(silu(gate) * up) from the fp32 split-K partials [range][rows][2n] of a gate_up GEMM
int launch_splitk_reduce_silu_mul(const float* partial, int ksplit, int rows, int n, uint16_t* out, int64_t out_stride,
                                  hipStream_t stream);
// split-K plan of a small-M fp8 dense GEMM on the mid kernel: 0 = shape not taken, else the number of K ranges (>= 1)
int mid_dense_ksplit(int M, int N, int K);
// m-tile table of a dense problem (one "expert"); identity_slots (optional, [M]) = 0..M-1 for the tuned kernels' row lookup
int launch_dense_tiles(int M, int tile_m, int4* tile_info, int* num_tiles, int* identity_slots, hipStream_t stream);
// per-row symmetric int8 quantisation: q = rint(x * 127/amax), scale = amax/127, amax = max(|row|, floor)
int launch_quant_int8_rows(const uint16_t* x, int64_t x_stride, int8_t* q, int64_t q_stride, float* scale, int64_t rows,
                           int cols, float floor, hipStream_t stream);

int launch_quant_int8_rows_f32(const float* x, int64_t x_stride, int8_t* q, int64_t q_stride, float* scale,
                               int64_t rows, int cols, float floor, hipStream_t stream);

// 32-token tiles, weights streamed global -> VGPR (moe_gemm_fp8w_stream.hip); tile table built with tile_m = 32
int launch_moe_gemm_fp8w_stream(int mode, const MoeGemmParams& p, int max_mtiles, hipStream_t stream);

// up to 128-token tiles, weights streamed global -> VGPR, activations K block by K block through LDS
// (moe_gemm_fp8w_mid.hip); tile table built with tile_m = 128
int launch_moe_gemm_fp8w_mid(int mode, const MoeGemmParams& p, int max_mtiles, hipStream_t stream);
// DOWN with two neighbouring column tiles per workgroup (p.n_tiles = output columns / 256; reduction % 256 == 0)
int launch_moe_gemm_fp8w_mid_down2(const MoeGemmParams& p, int max_mtiles, hipStream_t stream);

int launch_rmsnorm_bf16_dual(void* out, int64_t out_stride, void* out2, int64_t out2_stride, const void* x, int64_t x_stride,
                             const void* weight, int64_t rows, int hidden, float eps, hipStream_t stream);

// native MX-fp4 GEMM (gemm_mxfp4.hip): fp4 weights as stored on the block-scaled matrix cores, activations as two e4m3 terms
size_t mxfp4_native_workspace_bytes(int M, int N, int K);
bool mxfp4_native_ok(int M, int N, int K, const void* x, int64_t x_stride, const void* wq, const void* out, int64_t out_stride);
int launch_gemm_mxfp4_native(const void* x, int64_t x_stride, const void* wq, const void* scales, int scale_packed, const float* bias,
                             void* out, int64_t out_stride, int M, int N, int K, void* workspace, hipStream_t stream);

// moe_align with a second tile table: an expert's last tile lands in tile_info_b when it has at most tail_max rows
// (tail_max <= 0: exactly sglk_moe_align)
int launch_moe_align_split(const int32_t* topk_ids, int32_t M, int32_t E, int32_t topk, int32_t tile_m, int32_t* sorted_slot,
                           int32_t* expert_off, int32_t* tile_info, int32_t* num_tiles, int32_t tail_max,
                           int32_t* tile_info_b, int32_t* num_tiles_b, void* workspace, size_t workspace_bytes, void* stream,
                           int32_t* zero16 = nullptr,   // zero16: 16 ints cleared by the same launches (tile tickets)
                           const struct SplitJob* job = nullptr, bool* job_taken = nullptr);   // job: rows of `hidden` split by extra
                                                                                               // workgroups of the placing launch (fp8_split.h);
                                                                                               // *job_taken = false: the caller launches it

// the routed experts' per-slot rows, to be summed (valid slots, ascending) and scaled inside another kernel's epilogue
struct MoeSlotAddend {
    const uint16_t* ic2;          // [M * topk][K] bf16, slot order
    const int32_t* topk_ids;      // [M][topk]; ids outside [0, E) contribute nothing
    int topk, E;
};
bool shared_expert_can_fold(const sglk_shared_expert_args* a);
int shared_expert_impl(const sglk_shared_expert_args* a, void* stream, const MoeSlotAddend* moe);

// router (grouped top-k, softmax or sigmoid + bias) + moe_align in ONE launch for decode-size batches (topk.hip)
bool route_align_ok(int M, int E, int topk);
int launch_route_align(const void* gating, int64_t gating_stride, int gating_type, const void* bias, float* topk_weights,
                       int32_t* topk_ids, int M, int E, int topk, int renormalize, int G, int topk_group, int tile_m,
                       int32_t* sorted_slot, int32_t* expert_off, int32_t* tile_info, int32_t* num_tiles, hipStream_t s);

// out[m] = sum over valid slots j (ascending) of ic2[m*topk + j], fp32 sum, one bf16 rounding
int launch_moe_combine(const uint16_t* ic2, const int32_t* topk_ids, uint16_t* out, int64_t out_stride, int M,
                       int K, int E, int topk, hipStream_t stream);

}  // namespace sglk
