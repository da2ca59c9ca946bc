// Developer / test knobs of the library, read from the environment ONCE (first use) and cached; sglk_reload_env()
// (include/sglk.h) re-reads them, which is how the parity tests switch paths inside one process.  None of them changes a
// result beyond the stated tolerances; the defaults are what ships.
#pragma once
#include <atomic>

#include "sglk_common.h"

namespace sglk {

struct Knobs {
    int moe_tile_m = 0;          // SGLK_MOE_TILE_M: force the fp8 grouped-GEMM tiling (32 / 96 / 128 / 256); 0 = by batch size
    int no_tuned_splitk = 0;     // SGLK_NO_TUNED_SPLITK: the 256-row fp8 tile kernel never splits K in dense mode (A/B)
    int shared_big_wgs = 64;     // SGLK_SHARED_BIG_WGS: shared expert from 1024 rows on stays on the weight-streaming kernels while the tile kernels' gate_up launch would have at most this many workgroups (0 = off)
    int shared_i8_mid_max = 1024;// SGLK_SHARED_I8_MID_MAX: int8 shared expert below this M runs on the weight-streaming int8 kernel (several 128-row tiles); 129 = round-2 policy
    int shared_mid_max = 1024;   // SGLK_SHARED_MID_MAX: fp8 shared expert below this M runs as split-K passes of the weight-streaming kernel
    int dense_mid_max = 2048;    // SGLK_DENSE_MID_MAX: dense GEMMs below this M may take the weight-streaming kernels when the 256-row kernel would have few workgroups (192 = round-1 policy)
    int mid_i8_hi = 44, mid_bf16_hi = 44;   // SGLK_MID_I8_HI / SGLK_MID_BF16_HI: rows per expert below which int8 / bf16 experts take their weight-streaming kernels
    int mid_lo = 8, mid_hi = 0;   // SGLK_MID_LO / SGLK_MID_HI: crossovers (average rows per expert) stream -> mid -> 256; mid_hi 0 = by the experts' size (pick_tile_m)
    bool force_generic = false;  // SGLK_FORCE_GENERIC: every GEMM on the generic engine
    bool no_i8_mid = false;      // SGLK_NO_I8_MID
    int dense_mid_wgs_bf16 = 60, dense_mid_wgs_fp8 = 96;   // SGLK_DENSE_MID_WGS_BF16 / _FP8: from 192 rows on the weight-streaming kernel is taken while the 256-row kernel would have at most this many workgroups
    int mid_dense_model = 2;     // SGLK_MID_DENSE_MODEL: split-K of the dense fp8 / int8 weight-streaming kernels: 2 = rounds model on CUs slots, 1 = on 2 x CUs slots, 0 = aim at 512 workgroups (before)
    int bf16_mid_target = 0;     // SGLK_BF16_MID_TARGET: 0 = split-K of the dense bf16 weight-streaming kernel above 64 rows by the rounds model; n = aim at n workgroups (512 = before; profiles/r03_ab_dense_129_1000.txt)
    int i8_dense_mid_wgs = 64;   // SGLK_I8_DENSE_MID_WGS: dense int8 above 128 rows stays on the weight-streaming kernel while the 256-row kernel would have at most this many workgroups, and always below 192 rows (0 = round-2 policy: up to 128 rows)
    bool no_bf16_mid = false;    // SGLK_NO_BF16_MID
    int tail_split = -1;         // SGLK_TAIL_SPLIT: 0 = off, 1 = on the caller's stream, unset = caller's aux stream if given
    int mid_down2 = -1;          // SGLK_MID_DOWN2: 0 = one column tile per workgroup
    bool bf16_w4 = false;        // SGLK_BF16_W4: four-wave form of the bf16 256-row kernel
    bool align_3pass = false;    // SGLK_ALIGN_3PASS: moe_align always as count / scan / scatter
    bool wide_n = false;         // SGLK_WIDE_N: wave layout 2(n) x 4(m) of the fp8 256-row kernel
    int persist = -1;            // SGLK_PERSIST: 0 = one workgroup per tile, 1 = persistent; unset = by reduction length
    int max_wgs = 0;             // SGLK_MAX_WGS: cap on the persistent launches' workgroups (tests: forces many tiles per
                                 //               workgroup on small problems); 0 = one per CU
    int dec_nt = -1;             // SGLK_DEC_NT: 1 / 0 = decode_attention's cache rows are read with / without the non-temporal policy (-1 = on)
    int w_nt = -1;               // SGLK_W_NT: 1 / 0 = decode-size MoE kernels read the expert weights non-temporal / not (-1 = from 32 routed rows on)
    int dec_fold = -1;           // SGLK_DEC_FOLD: 0 = decode_attention's cache write always as its own launch (-1 = folded when small)
    int dec_splits = 0;          // SGLK_DEC_SPLITS: KV splits decode_attention uses (0 = one round of workgroups, -1 = all the scratch has)
    int attn_nw = 0;             // SGLK_ATTN_NW: waves per extend-attention workgroup (4 / 8); 0 = by launch size
    int attn_pair = -1;          // SGLK_ATTN_PAIR: heavy + light causal query blocks in one workgroup (0 / 1); unset = by launch size
    int attn_pp = -1;            // SGLK_ATTN_PP: 0 = one-phase extend kernel for D = DV = 128 too (A/B)
    int pack_min_rows = 0;       // SGLK_PACK_MIN_ROWS: row-major dense weights are re-tiled into the workspace from this many rows on (below: generic engine); 0 = by weight type and layer size (gemm_api.hip: pack_on_the_fly), 192 = before
    int no_pack_on_the_fly = 0;  // SGLK_NO_PACK_ON_THE_FLY: row-major dense weights stay on the generic engine at every M (A/B)
    int mxfp4_rt = 4;            // SGLK_MXFP4_RT: 32-row tiles per wave of the fp4-MFMA kernel (4 / 2; A/B)
    int mxfp4_native = -1;       // SGLK_MXFP4_NATIVE: 1 / 0 = fp4-MFMA kernel for every legal shape / never; unset = by tile count
    int attn_order = -1;         // SGLK_ATTN_ORDER: A/B override of the extend-attention dispatch order
    int inline_align_max = 16;   // SGLK_INLINE_ALIGN_MAX: fused_experts with at most this many slots (<= 32) sorts the ids in the GEMM kernels, no moe_align launch; 0 = never (A/B)
    bool no_block_fold = false;  // SGLK_NO_BLOCK_FOLD: sglk_moe_block runs router / align / combine / shared expert unfused (A/B)
    int s128 = -1;               // SGLK_S128: 1 / 0 = the split on 128-token tiles with two workgroups per CU (moe_gemm_fp8w_s128.hip) / never; unset = default
    int ar_wait_ms = 0;          // SGLK_AR_WAIT_MS: how long the direct all-reduce waits for a peer before it gives up (default 30000)
    int dense_s128 = -1;         // SGLK_DENSE_S128: 0 = dense fp8 / int8 GEMMs never on the 128-token kernel; 1 = from 128 rows; unset = from 1024 of its tiles
    int mid_nw = 0, mid_far = -1; // SGLK_MID_NW (4 / 8), SGLK_MID_FAR (0 / 1): force the mid kernel's GEMM-1 workgroup width / far activation prefetch (A/B)
    bool no_mid_narrow = false;  // SGLK_NO_MID_NARROW: the mid kernel's GEMM-1 keeps 128 columns per workgroup at decode sizes too (A/B)
    int s128_prio = -1;          // SGLK_S128_PRIO: 0 = no wave priority in the 128-token kernel, 1 = s_setprio 1 inside its main loop, 2 = inside its prologue / epilogue; unset = 1 for GEMM-2 only
    int i8_s128 = -1;            // SGLK_I8_S128: 1 / 0 = large-M int8 W8A8 fused_experts on the 128-token kernel (terms = 0) / on gemm_i8_256.hip
    int fp8_act = 0;             // SGLK_FP8_ACT: 1 = opt-in a8 mode (fp8 activations on the block-scaled fp8 matrix cores)
    int rescale_ablate = 0;      // SGLK_RESCALE (SGLK_DEV_ABLATE builds only)
    unsigned long long dbg_ptr = 0;   // SGLK_DBG_PTR (SGLK_DEV_ABLATE builds only)
};

const Knobs& knobs();

// compute units of the CURRENT device (cached per device)
int device_cu_count();
// split-K by the rounds model (weight-streaming dense kernels): the launch runs ceil(tiles x ranges / slots) rounds, each as long as one
// range plus ~3 K blocks' worth of prologue and epilogue; a split adds the reduce launch (~3 blocks) and the fp32 / int32 partials' round
// trip (~7 MB per block time).  Returns the range count with the lowest cost (0 = none allowed).  Fitted to same-box A/Bs at 1 ... 1000
// rows x five layer shapes (tools/ab_mid_dense_model.py, tools/ab_bf16_mid_target.py; profiles/r03_ab_dense_129_1000.txt)
inline int splitk_by_rounds(int kblocks, long long tiles, int M, int N, int slots, int per_cap = 1 << 30) {
    const int per_min = kblocks >= 4 ? 4 : 2;
    int best = 0;
    long long best_cost = -1;
    for (int per = kblocks; per >= per_min; per -= 2) {
        if (kblocks % per != 0 || per > per_cap) continue;
        const int ks = kblocks / per;
        if (ks > 32 || (long long)ks * M * N * 4 > (64ll << 20)) continue;
        long long cost = ((tiles * ks + slots - 1) / slots) * (per + 3) * 16;   // sixteenths of a block time
        if (ks > 1) cost += 48 + (long long)ks * M * N * 128 / 7000000;
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = ks;
        }
    }
    return best;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (call site, device); `done` is the call site's bit mask
int ensure_dyn_lds(const void* func, int bytes, std::atomic<unsigned>& done, const char* what);

#define SGLK_ENSURE_DYN_LDS(func, bytes, what)                                            \
    do {                                                                                  \
        static std::atomic<unsigned> done_{0};                                            \
        const int rc_ = ::sglk::ensure_dyn_lds((const void*)(func), (int)(bytes), done_, what); \
        if (rc_ != SGLK_OK) return rc_;                                                   \
    } while (0)

}  // namespace sglk
