// fused_experts: the C-ABI entry point that chains align -> GEMM-1(+SiLU*mul) -> GEMM-2(*topk_w) -> combine.
// Operator contract: /root/reference/bench_moe.py:113-130 (14-arg), /root/reference/test_moe.py:79-92 (13-arg).
#include "knobs.h"
#include "moe_internal.h"
#include "moe_align_inline.h"
#include "fp8_split.h"

#include <stdlib.h>

#include <vector>

using namespace sglk;

namespace {

constexpr bool kS128Default = true;
constexpr bool kI8S128Default = true;   // large-M int8 W8A8 on the 128-token kernel (exact int32 sums, ic1 quantised in GEMM-1's epilogue) when SGLK_I8_S128 is unset
    // large-M W8A16 on the 128-token two-workgroups-per-CU split kernel when SGLK_S128 is unset

struct StageTimer {
    int max_calls = 0, calls = 0;
    std::vector<hipEvent_t> ev;   // (SGLK_NUM_STAGES + 1) events per call
    hipEvent_t at(int call, int i) { return ev[(size_t)call * (SGLK_NUM_STAGES + 1) + i]; }
};

struct Workspace {
    size_t align_ws, sorted_slot, expert_off, tile_info, num_tiles, tile_info_b, num_tiles_b, tickets, ic1, ic2, xq, xs, ic1q, ic1s, i8_sync, i8_sync_bytes, wpack, total;
};

// Tile height of the tuned grouped GEMMs, from the average rows an expert receives (S/E).  Measured crossovers at Qwen3
// dims (tools/moe_stage_sweep.py, profiles/r01_v7_stage_sweep.txt):
//   <   8 rows : 32-token tiles, weight-streaming kernel, activations whole in LDS (decode: M < 128)
//   <  72 rows : up to 96-token tiles, weight-streaming kernel with K-blocked activations (M = 128 .. 1151): every touched
//                expert's weights are read ONCE (a 32-token tile reads them 2-3 times, a 256-token tile wastes its ring
//                and half its waves on padding); GEMM-1 runs at 4.5-5.5 TB/s of weight stream
//   >= 72 rows : 256-token tiles, 8-wave LDS-DMA ring kernel
//   shapes these cannot take (K % 256, very large operands): 128-token tiles
int pick_tile_m(int M, int N, int K, int E, int topk, int block_n, int64_t hidden_stride) {
    const Knobs& kn = knobs();
    const int64_t S = (int64_t)M * topk;
    // the 256 kernel addresses its operands through 32-bit buffer offsets (the rows of `hidden` by their STRIDE, which a
    // row-strided view makes larger than K), walks the scale table in 32-row operand tiles and needs two K blocks per GEMM
    const bool ok256 = (K % 256 == 0) && (N % 128 == 0) && N >= 256 && block_n % 32 == 0 && S * (int64_t)N * 2 < (1ll << 32) &&
                       (int64_t)M * hidden_stride * 2 < (1ll << 32) && (int64_t)2 * N * K < (1ll << 32);
    // stream kernel: both reduction lengths (K for GEMM-1, N for GEMM-2) must be multiples of 256 (ring of 8 pieces)
    const bool ok_stream = (K % 256 == 0) && (N % 256 == 0) && (int64_t)kStreamTileM * K * 2 <= 150 * 1024 &&
                           (int64_t)kStreamTileM * N * 2 <= 150 * 1024;
    // mid kernel: reduction lengths in whole 128s, 2 .. 64 blocks (scale table), 128 ic1 / 128 output columns per workgroup
    const bool ok_mid = (K % 128 == 0) && (N % 128 == 0) && K >= 256 && N >= 256 && K <= 8192 && N <= 8192;
    if (kn.moe_tile_m > 0) {
        const int f = kn.moe_tile_m;
        if (f == 256 && ok256) return 256;
        if (f == 32 && ok_stream) return kStreamTileM;
        if (f == kMidTileM && ok_mid) return kMidTileM;
        if (f == 128) return 128;
    }
    // crossovers in average rows per expert (A/B overrides SGLK_MID_LO / SGLK_MID_HI).  mid -> 256: above 96 rows an expert needs
    // two mid tiles and streams its weights twice; at Qwen3's shape (N = 768, K = 2048) and at DeepSeek-like experts (N = 2048,
    // K = 7168) that is a wash or a small win up to ~160 rows.  Narrow experts behind a deep reduction are different: the mid
    // kernel's DOWN stage is thousands of workgroups with a three-block reduction each, and the 256-row kernel wins from ~72 rows on
    // (tools/ab_moe_shapes.py, profiles/r02_ab_moe_shapes.txt: N = 384, K = 7168, E = 256 at 123 rows per expert -- the reference
    // bench's own shape, bench_moe.py:144-145 -- 1.14 -> 0.94 ms)
    // With the 128-token two-workgroups-per-CU kernel behind the "256" answer (moe_gemm_fp8w_s128.hip: one tile per expert at 128
    // rows, token tiles without rows skipped) the crossover sits at ~64 rows per expert: M = 1024 / 1536 / 2048 at Qwen3 dims
    // 419 -> 432 / 464 -> 559 / 519 -> 591 TFLOP/s (profiles/r03_ab_mid_vs_s128.txt)
    const bool s128_next = ok256 && moe_gemm_fp8w_s128_ok(N, K, block_n) && (kn.s128 >= 0 ? kn.s128 == 1 : kS128Default) &&
                           (int64_t)M * K * 2 < (1ll << 32) && S * (int64_t)N * 2 < (1ll << 32);
    const int64_t lo = kn.mid_lo, hi = kn.mid_hi > 0 ? kn.mid_hi : (s128_next ? 64 : ((N <= 512 && K >= 4096) ? 72 : 160));
    if (ok_stream && S < lo * E) return kStreamTileM;
    if (ok_mid && S < hi * E) return kMidTileM;
    if (ok_stream && !ok_mid && S < (int64_t)44 * E) return kStreamTileM;
    return ok256 ? 256 : kTileM;
}

Workspace plan_workspace(int M, int N, int K, int E, int topk, int wtype, int flags) {
    Workspace w{};
    const int64_t S = (int64_t)M * topk;
    const int max_tiles = sglk_moe_max_tiles(M, E, topk, kStreamTileM);   // the smallest tile bounds the table size
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes ? bytes : 1, 256);
        return o;
    };
    w.align_ws = take(sglk_moe_align_workspace_bytes(M, E, topk));
    w.sorted_slot = take((size_t)S * sizeof(int));
    w.expert_off = take((size_t)(E + 1) * sizeof(int));
    w.tile_info = take((size_t)max_tiles * 4 * sizeof(int));
    w.num_tiles = take(sizeof(int));
    w.tile_info_b = take((size_t)E * 4 * sizeof(int));   // tail tiles (at most one per expert) of the 256-row plan
    w.num_tiles_b = take(sizeof(int));
    w.tickets = take(16 * sizeof(int));
    w.ic1 = take((size_t)S * N * (wtype == SGLK_W_INT8 ? 4 : 2));   // W8A8 keeps SiLU*mul in fp32 until it is quantised
    w.ic2 = take((size_t)S * K * 2);
    if (wtype == SGLK_W_FP8_E4M3 && !(flags & SGLK_MOE_FP8_ACT)) {   // two-term split of `hidden` (ic1's split rows take ic1's own place)
        w.xq = take((size_t)M * 2 * K);
        w.xs = take((size_t)M * align_up(K / 128, 4));
        w.ic1s = take((size_t)S * align_up(N / 128, 4));
    }
    if (wtype == SGLK_W_FP8_E4M3 && (flags & SGLK_MOE_FP8_ACT)) {   // a8 mode: e4m3 copies + one e8m0 scale per 128-wide block
        w.xq = take((size_t)M * K);
        w.xs = take((size_t)M * align_up(K / 128, 4));
        w.ic1q = take((size_t)S * N);
        w.ic1s = take((size_t)S * align_up(N / 128, 4));
    }
    if (wtype == SGLK_W_INT8) {   // W8A8: dynamically quantised activations of both GEMMs
        w.xq = take((size_t)M * K);
        w.xs = take((size_t)M * sizeof(float));
        w.ic1q = take((size_t)S * N);
        w.ic1s = take((size_t)S * sizeof(float));
        // 128-token kernel: per-row maxima of silu(gate) * up + one arrival counter per m-tile (zeroed in front of GEMM-1)
        w.i8_sync_bytes = (size_t)S * sizeof(unsigned) + (size_t)sglk_moe_max_tiles(M, E, topk, kTileM) * sizeof(int);
        w.i8_sync = take(w.i8_sync_bytes);
    }
    if (flags & SGLK_MOE_PACK_WEIGHTS)   // re-tiled copies of row-major w1 and w2
        w.wpack = take((size_t)E * 3 * N * K * (wtype == SGLK_W_BF16 ? 2 : 1));
    w.total = off;
    return w;
}

// The tail tiles' GEMM-1 -> GEMM-2 chain is independent of the full tiles' (it reads and writes its own rows of ic1 / ic2),
// so it can run beside the two big launches and fill the CUs their last tiles leave idle: fork after moe_align, join before
// the combine.  The second stream and the two events are the CALLER's (args->aux_stream / aux_events, e.g. from
// sglk_aux_create); the library creates and keeps nothing.

// shapes / layouts the tuned fp8 kernels (moe_gemm_fp8w*.hip) accept; everything else runs on the generic engine
bool tuned_fp8_ok(const sglk_fused_experts_args* a) {
    return a->wtype == SGLK_W_FP8_E4M3 && (a->packed & 3) == 3 && a->N % 128 == 0 && a->K % 128 == 0 && a->block_k == 128 &&
           a->block_n > 0 && a->block_n % 16 == 0 && (a->hidden_stride % 8) == 0 && ((uintptr_t)a->hidden % 16) == 0 &&
           !knobs().force_generic;
}

// int8 fused_experts on the int8 matrix cores (gemm_i8_256.hip): packed weights, both reduction lengths whole 64-deep
// stages (>= 4 of them), output tiles whole, and enough rows per expert for 256-token tiles to pay (same threshold as the
// fp8 path).  32-bit buffer offsets bound the operand sizes.
// int8 experts below the 256-row kernel's range: weight-streaming int8 kernel (gemm_i8_mid.hip), tiles of up to 128 rows
bool mid_int8_ok(const sglk_fused_experts_args* a) {
    const int64_t S = (int64_t)a->M * a->topk;
    return a->wtype == SGLK_W_INT8 && (a->packed & 3) == 3 && a->K % 128 == 0 && a->N % 128 == 0 && a->K >= 256 && a->N >= 256 &&
           S < (int64_t)knobs().mid_i8_hi * a->E && !knobs().force_generic && !knobs().no_i8_mid;
}

bool tuned_int8_ok(const sglk_fused_experts_args* a) {
    const int64_t S = (int64_t)a->M * a->topk;
    return a->wtype == SGLK_W_INT8 && (a->packed & 3) == 3 && a->K % 256 == 0 && a->N % 128 == 0 && a->N >= 256 &&
           S >= (int64_t)knobs().mid_i8_hi * a->E && (int64_t)a->M * a->K < (1ll << 32) && S * (int64_t)a->N < (1ll << 32) &&
           (int64_t)2 * a->N * a->K < (1ll << 32) && !knobs().force_generic;
}

// bf16 fused_experts on the tuned bf16 kernel (gemm_bf16_256.hip): VNNI-2 packed weights, whole tiles, large M
// bf16 experts below the 256-row kernel's range (< 44 rows per expert; at 64 rows the two measured the same, both bound by
// the 1.2 GB of bf16 weights): weight-streaming kernel on the packed order (gemm_bf16_mid.hip), tiles of up to 96 rows
bool mid_bf16_ok(const sglk_fused_experts_args* a) {
    const int64_t S = (int64_t)a->M * a->topk;
    return a->wtype == SGLK_W_BF16 && (a->packed & 3) == 3 && a->K % 128 == 0 && a->N % 128 == 0 && a->K >= 256 && a->N >= 256 &&
           S < (int64_t)knobs().mid_bf16_hi * a->E && a->hidden_stride % 8 == 0 && ((uintptr_t)a->hidden % 16) == 0 &&
           !knobs().force_generic && !knobs().no_bf16_mid;
}

bool tuned_bf16_ok(const sglk_fused_experts_args* a) {
    const int64_t S = (int64_t)a->M * a->topk;
    return a->wtype == SGLK_W_BF16 && (a->packed & 3) == 3 && a->K % 256 == 0 && a->N % 128 == 0 && a->N >= 128 &&
           S >= (int64_t)knobs().mid_bf16_hi * a->E && a->hidden_stride % 8 == 0 && ((uintptr_t)a->hidden % 16) == 0 &&
           (int64_t)a->M * a->hidden_stride * 2 < (1ll << 32) && S * (int64_t)a->N * 2 < (1ll << 32) &&
           (int64_t)4 * a->N * a->K < (1ll << 32) && !knobs().force_generic;
}

}  // namespace

extern "C" size_t sglk_fused_experts_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t E, int32_t topk,
                                                     int32_t wtype) {
    if (M < 0 || N <= 0 || K <= 0 || E <= 0 || topk <= 0) return 0;
    return plan_workspace(M, N, K, E, topk, wtype, 0).total;
}

extern "C" size_t sglk_fused_experts_workspace_bytes_ex(int32_t M, int32_t N, int32_t K, int32_t E, int32_t topk,
                                                        int32_t wtype, int32_t flags) {
    if (M < 0 || N <= 0 || K <= 0 || E <= 0 || topk <= 0) return 0;
    return plan_workspace(M, N, K, E, topk, wtype, flags).total;
}

namespace {
// router inputs of sglk_moe_block: topk_weights / topk_ids of the experts' args are then OUTPUTS of the call
struct RouteArgs {
    const void* gating;
    int64_t gating_stride;
    int gating_type;
    const void* bias;
    int renormalize, G, topk_group;
};
int fused_experts_impl(const sglk_fused_experts_args* a, void* stream, const RouteArgs* route, const sglk_shared_expert_args* shared);
}  // namespace

extern "C" int sglk_fused_experts(const sglk_fused_experts_args* a, void* stream) { return fused_experts_impl(a, stream, nullptr, nullptr); }

namespace {
int fused_experts_impl(const sglk_fused_experts_args* a, void* stream, const RouteArgs* route, const sglk_shared_expert_args* shared) {
    SGLK_REQUIRE(a, SGLK_ERR_INVALID, "fused_experts: null args");
    const int M = a->M, N = a->N, K = a->K, E = a->E, topk = a->topk;
    SGLK_REQUIRE(M >= 0 && N > 0 && K > 0 && E > 0 && topk > 0, SGLK_ERR_INVALID,
                 "fused_experts: bad sizes M=%d N=%d K=%d E=%d topk=%d", M, N, K, E, topk);
    SGLK_REQUIRE(a->w1 && a->w2 && a->workspace, SGLK_ERR_INVALID, "fused_experts: null pointer");
    SGLK_REQUIRE(M == 0 || (a->hidden && a->out && a->topk_weights && a->topk_ids), SGLK_ERR_INVALID,
                 "fused_experts: null pointer");
    SGLK_REQUIRE(a->hidden_stride >= K && a->out_stride >= K, SGLK_ERR_INVALID, "fused_experts: row stride < K");
    SGLK_REQUIRE((int64_t)M * topk * (int64_t)(K > N ? K : N) < (1ll << 40), SGLK_ERR_SHAPE, "fused_experts: too large");
    SGLK_REQUIRE(a->wtype == SGLK_W_BF16 || a->wtype == SGLK_W_FP8_E4M3 || a->wtype == SGLK_W_INT8, SGLK_ERR_INVALID,
                 "fused_experts: unknown weight type %d", a->wtype);
    hipStream_t s = (hipStream_t)stream;

    if (a->wtype == SGLK_W_FP8_E4M3) {
        SGLK_REQUIRE(a->w1_scale && a->w2_scale, SGLK_ERR_INVALID, "fused_experts: fp8 needs w1_scale and w2_scale");
        SGLK_REQUIRE(a->block_k == 128, SGLK_ERR_SHAPE, "fused_experts: block_size[1] must be 128 (got %d)", a->block_k);
        SGLK_REQUIRE(a->block_n > 0 && a->block_n % 16 == 0, SGLK_ERR_SHAPE,
                     "fused_experts: block_size[0] must be a positive multiple of 16 (got %d)", a->block_n);
        SGLK_REQUIRE(N % 16 == 0, SGLK_ERR_SHAPE, "fused_experts(fp8): N (%d) must be a multiple of 16", N);
    } else if (a->wtype == SGLK_W_INT8) {
        SGLK_REQUIRE(a->w1_scale && a->w2_scale, SGLK_ERR_INVALID, "fused_experts: int8 needs w1_scale [E,2N] and w2_scale [E,K]");
    }
    if ((a->flags & SGLK_MOE_PACK_WEIGHTS) && a->packed == 0 && M > 0) {
        const bool ok1 = a->wtype == SGLK_W_BF16 ? ((2 * N) % 32 == 0 && K % 8 == 0) : ((2 * N) % 16 == 0 && K % 64 == 0);
        const bool ok2 = a->wtype == SGLK_W_BF16 ? (K % 32 == 0 && N % 8 == 0) : (K % 16 == 0 && N % 64 == 0);
        const Workspace wp = plan_workspace(M, N, K, E, topk, a->wtype, a->flags);
        if (ok1 && ok2 && a->workspace_bytes >= wp.total) {
            const size_t elt = a->wtype == SGLK_W_BF16 ? 2 : 1;
            unsigned char* p1 = (unsigned char*)a->workspace + wp.wpack;
            unsigned char* p2 = p1 + (size_t)E * 2 * N * K * elt;
            int rcp = sglk_pack_weight(a->w1, p1, E, 2 * N, K, a->wtype, stream);
            if (rcp != SGLK_OK) return rcp;
            rcp = sglk_pack_weight(a->w2, p2, E, K, N, a->wtype, stream);
            if (rcp != SGLK_OK) return rcp;
            sglk_fused_experts_args b = *a;
            b.w1 = p1;
            b.w2 = p2;
            b.packed = 3;
            b.flags &= ~SGLK_MOE_PACK_WEIGHTS;
            return fused_experts_impl(&b, stream, route, shared);
        }
    }
    if (a->packed & 1) {
        const bool ok = a->wtype == SGLK_W_BF16 ? ((2 * N) % 32 == 0 && K % 8 == 0) : ((2 * N) % 16 == 0 && K % 64 == 0);
        SGLK_REQUIRE(ok, SGLK_ERR_SHAPE, "fused_experts: w1 [%d][%d] cannot be in packed order; clear packed bit 0", 2 * N, K);
    }
    if (a->packed & 2) {
        const bool ok = a->wtype == SGLK_W_BF16 ? (K % 32 == 0 && N % 8 == 0) : (K % 16 == 0 && N % 64 == 0);
        SGLK_REQUIRE(ok, SGLK_ERR_SHAPE, "fused_experts: w2 [%d][%d] cannot be in packed order; clear packed bit 1", K, N);
    }
    SGLK_REQUIRE(((uintptr_t)a->out % 2) == 0 && ((uintptr_t)a->hidden % 2) == 0, SGLK_ERR_INVALID, "fused_experts: misaligned");

    // SGLK_MOE_PACK_WEIGHTS is a hint: when the re-tiled copy was not made above (weights already packed, M == 0, a shape the
    // tuned kernels do not take, a workspace sized without the flag) the call runs on the weights as given
    SGLK_REQUIRE((a->flags & ~(SGLK_MOE_FP8_ACT | SGLK_MOE_PACK_WEIGHTS)) == 0, SGLK_ERR_INVALID, "fused_experts: unknown flags 0x%x", a->flags);
    const Workspace w = plan_workspace(M, N, K, E, topk, a->wtype, a->flags & ~SGLK_MOE_PACK_WEIGHTS);   // (the copy is the plan's last region)
    SGLK_REQUIRE(a->workspace_bytes >= w.total, SGLK_ERR_WORKSPACE, "fused_experts: workspace %zu < required %zu",
                 a->workspace_bytes, w.total);
    SGLK_REQUIRE(((uintptr_t)a->workspace % 256) == 0, SGLK_ERR_INVALID, "fused_experts: workspace must be 256-B aligned");
    if (M == 0) return SGLK_OK;

    unsigned char* ws = (unsigned char*)a->workspace;
    int* sorted_slot = (int*)(ws + w.sorted_slot);
    int* expert_off = (int*)(ws + w.expert_off);
    int* tile_info = (int*)(ws + w.tile_info);
    int* num_tiles = (int*)(ws + w.num_tiles);
    uint16_t* ic1 = (uint16_t*)(ws + w.ic1);
    uint16_t* ic2 = (uint16_t*)(ws + w.ic2);

    StageTimer* tm = (StageTimer*)a->stage_timer;
    if (tm && tm->calls >= tm->max_calls) tm = nullptr;   // pool exhausted: stop recording
    const int call = tm ? tm->calls++ : 0;
    auto mark = [&](int i) {
        if (tm) hipEventRecord(tm->at(call, i), s);
    };
    const bool tuned = tuned_fp8_ok(a);
    const bool a8 = (a->flags & SGLK_MOE_FP8_ACT) != 0;
    if (a8) {   // an explicit request: refuse what the a8 kernels cannot take instead of answering with other numerics
        SGLK_REQUIRE(tuned && K % 256 == 0 && N % 128 == 0 && N >= 256 && a->block_n % 32 == 0 && K <= 8192 && N <= 8192 &&
                         (int64_t)M * K < (1ll << 32) && (int64_t)M * topk * N < (1ll << 32) && (int64_t)2 * N * K < (1ll << 32),
                     SGLK_ERR_SHAPE, "fused_experts: SGLK_MOE_FP8_ACT needs packed fp8 weights, block [32k,128], K %% 256 == 0, "
                     "N %% 128 == 0, K,N <= 8192 (got N=%d K=%d block_n=%d packed=%d)", N, K, a->block_n, a->packed);
    }
    const bool mid_i8 = mid_int8_ok(a);
    const bool tuned_i8 = !mid_i8 && tuned_int8_ok(a);
    const bool mid_b16 = mid_bf16_ok(a);
    const bool tuned_b16 = !mid_b16 && tuned_bf16_ok(a);
    int tile_m = a8 ? 256 : tuned ? pick_tile_m(M, N, K, E, topk, a->block_n, a->hidden_stride) : (mid_b16 ? kMidTileM : (mid_i8 ? kI8MidTileM : ((tuned_i8 || tuned_b16) ? 256 : kGenericTileM)));
    // the 256-row regime on 128-token tiles, four waves, two workgroups per CU (moe_gemm_fp8w_s128.hip): the two-term split
    // whose prologue / epilogue hide behind the co-resident workgroup's main loop
    // (the default only when no test knob pins a tiling or one of the 256-row kernels; SGLK_S128=1 / 0 forces / forbids it)
    const bool s128_on = knobs().s128 >= 0 ? knobs().s128 == 1 : (kS128Default && knobs().moe_tile_m == 0);
    const bool s128 = tuned && !a8 && tile_m == 256 && moe_gemm_fp8w_s128_ok(N, K, a->block_n) && (int64_t)M * K * 2 < (1ll << 32) &&
                      (int64_t)M * topk * N * 2 < (1ll << 32) && s128_on;
    if (s128) tile_m = 128;
    const bool a8s = a8;   // (the a8 request was checked against moe_gemm_fp8w_s128_ok's shape rules above)
    if (a8s) tile_m = 128;
    // (an m-tile's N / 128 GEMM-1 workgroups wait for each other in the epilogue: with many column tiles the first ones would hold
    // their slots for a good part of a tile's time, so wide experts stay on the 256-row kernels + separate quantisation pass)
    const bool i8s = tuned_i8 && moe_gemm_fp8w_s128_ok(N, K, 32) && N / 128 <= 16 && (knobs().i8_s128 >= 0 ? knobs().i8_s128 == 1 : kI8S128Default);
    if (i8s) tile_m = 128;
    // 256-row plan: the last of an expert's several tiles, when it has at most 96 rows, is taken out of the table and run on
    // the weight-streaming mid kernel, where it costs what its rows cost instead of a whole 256-row tile (M = 4096: 61 of 189
    // tiles).  SGLK_TAIL_SPLIT=0 switches it off.
    // SGLK_TAIL_SPLIT: 0 = off, 1 = on the caller's stream; unset = on the caller's aux stream when one is given, else off
    // (on the caller's own stream the tails cost more than they save: 0.454 vs 0.440 ms at M = 4096, 0.402 beside)
    const int tail_knob = knobs().tail_split;
    const bool have_aux = a->aux_stream && a->aux_events[0] && a->aux_events[1];
    // Same-box A/B at Qwen3 dims (tools/ab_tail_split.sh): M = 3929 -1.8 %, 4096 -8 %, 8192 -0.7 %, but 16384 +1.5 % and
    // 32768 +1.7 % (few tails per full tile, and the side launches get in the big kernels' way) -> only below ~640 rows per
    // expert.
    const bool split_tails = tuned && tile_m == 256 && K % 256 == 0 && N % 256 == 0 && K <= 4096 && N <= 4096 &&
                             (int64_t)M * topk < (int64_t)640 * E && !(a->flags & SGLK_MOE_FP8_ACT) &&
                             (tail_knob == 1 || tail_knob == 2 || (tail_knob != 0 && have_aux));
    const bool side = split_tails && tail_knob != 1 && tail_knob != 2 && have_aux;
    int* tile_info_b = (int*)(ws + w.tile_info_b);
    int* num_tiles_b = (int*)(ws + w.num_tiles_b);
    mark(0);
    int rc = SGLK_OK;
    bool routed_and_aligned = false;
    if (route) {
        // router: grouped top-k writes topk_weights / topk_ids; decode-size batches do it in the SAME launch as the align
        if (route_align_ok(M, E, topk) && !split_tails && !knobs().no_block_fold) {
            rc = launch_route_align(route->gating, route->gating_stride, route->gating_type, route->bias, (float*)a->topk_weights,
                                    (int32_t*)a->topk_ids, M, E, topk, route->renormalize, route->G, route->topk_group, tile_m,
                                    sorted_slot, expert_off, tile_info, num_tiles, s);
            routed_and_aligned = true;
        } else {
            rc = sglk_grouped_topk(route->gating, route->gating_stride, route->gating_type, route->bias, (float*)a->topk_weights,
                                   (int32_t*)a->topk_ids, M, E, topk, route->renormalize, route->G, route->topk_group, stream);
        }
        if (rc != SGLK_OK) return rc;
    }
    // decode sizes on the weight-streaming kernel: no align launch at all -- its workgroups sort the (at most 32) ids themselves
    // (moe_align_inline.h).  Four dependent launches become three: M = 1 (8 slots) 33.7 -> 30.5 us as a hipGraph replay; at 32 slots
    // the sort costs what the launch did, hence 16 (SGLK_INLINE_ALIGN_MAX, 0 = off).
    const int inline_max = knobs().inline_align_max < kInlineAlignSlots ? knobs().inline_align_max : kInlineAlignSlots;
    const bool inline_align = tuned && !a8 && tile_m == kStreamTileM && !route && !split_tails && (int64_t)M * topk <= inline_max;
    // the two-term split of `hidden` (W8A16 on the scaled fp8 MFMA) rides in moe_align's second launch as extra workgroups
    const bool want_split = s128;
    SplitJob sjob{};
    bool split_done = false;
    if (want_split) {
        sjob.x = (const uint16_t*)a->hidden;
        sjob.x_stride = a->hidden_stride;
        sjob.q = ws + w.xq;
        sjob.q_stride = 2 * (int64_t)K;
        sjob.s = ws + w.xs;
        sjob.s_stride = (int64_t)align_up(K / 128, 4);
        sjob.rows = M;
        sjob.cols = K;
        sjob.terms = 2;
    }
    if (a8) {   // the a8 mode's quantisation pass rides the same way
        sjob.x = (const uint16_t*)a->hidden;
        sjob.x_stride = a->hidden_stride;
        sjob.q = ws + w.xq;
        sjob.q_stride = K;
        sjob.s = ws + w.xs;
        sjob.s_stride = (int64_t)align_up(K / 128, 4);
        sjob.rows = M;
        sjob.cols = K;
        sjob.terms = 1;
    }
    if (i8s) {   // the per-token int8 rows of `hidden` ride the same way
        sjob.x = (const uint16_t*)a->hidden;
        sjob.x_stride = a->hidden_stride;
        sjob.q = ws + w.xq;
        sjob.q_stride = K;
        sjob.sf = (float*)(ws + w.xs);
        sjob.floor_v = 1e-7f;
        sjob.rows = M;
        sjob.cols = K;
        sjob.terms = 0;
    }
    // the one-launch router + align does not clear the persistent kernels' tile tickets (moe_align's launches do)
    if (routed_and_aligned && tile_m == 256 && hipMemsetAsync(ws + w.tickets, 0, 16 * sizeof(int), s) != hipSuccess)
        SGLK_FAIL(SGLK_ERR_LAUNCH, "fused_experts: ticket reset failed");
    if (!routed_and_aligned && !inline_align)
        rc = launch_moe_align_split(a->topk_ids, M, E, topk, tile_m, sorted_slot, expert_off, tile_info, num_tiles,
                                    split_tails ? kMidTileM : 0, tile_info_b, num_tiles_b, ws + w.align_ws,
                                    w.sorted_slot - w.align_ws, stream, (int32_t*)(ws + w.tickets), (want_split || a8 || i8s) ? &sjob : nullptr,
                                    &split_done);
    if (rc != SGLK_OK) return rc;
    mark(1);
    const int max_tiles = sglk_moe_max_tiles(M, E, topk, tile_m);

    // large batches: the two-term e4m3 split on the scaled fp8 matrix cores, 128-token tiles (moe_gemm_fp8w_s128.hip) -- the same
    // W8A16 contract as the bf16-MFMA kernels below (SGLK_S128=0 / 1 overrides)
    const bool split = want_split;
    if (split) {
        uint8_t* xq = ws + w.xq;
        uint8_t* xs = ws + w.xs;
        uint8_t* ic1q = (uint8_t*)ic1;                 // split rows [position][2N] bytes: exactly the bf16 ic1's footprint
        uint8_t* ic1s = ws + w.ic1s;
        const int xs_stride = (int)align_up(K / 128, 4), ic1s_stride = (int)align_up(N / 128, 4);
        if (!split_done) {   // (the one-launch align of small inputs has no second kernel to ride in)
            rc = launch_split_fp8_block128((const uint16_t*)a->hidden, a->hidden_stride, xq, 2 * (int64_t)K, xs, xs_stride, M, K, s);
            if (rc != SGLK_OK) return rc;
            mark(1);   // the split pass counts towards the align stage
        }
        A8GemmParams q1{};
        q1.x = xq;
        q1.x_stride = 2 * (int64_t)K;
        q1.x_bytes = (int64_t)M * 2 * K;
        q1.xs = xs;
        q1.xs_stride = xs_stride;
        q1.sorted_slot = sorted_slot;
        q1.topk = topk;
        q1.w = (const uint8_t*)a->w1;
        q1.w_expert_stride = (int64_t)2 * N * K;
        q1.w_scale = a->w1_scale;
        q1.scale_rows = (int)ceil_div(2 * N, a->block_n);
        q1.scale_cols = K / 128;
        q1.block_n = a->block_n;
        q1.C = K;
        q1.n_half = N;
        q1.tile_info = (const int4*)tile_info;
        q1.num_tiles = num_tiles;
        q1.n_tiles = N / 128;
        q1.out = ic1q;
        q1.out_stride = 2 * (int64_t)N;
        q1.out_s = ic1s;
        q1.out_s_stride = ic1s_stride;
        q1.max_mtiles = max_tiles;
#ifdef SGLK_DEV_ABLATE
        if (knobs().dbg_ptr) q1.dbg = (unsigned long long*)knobs().dbg_ptr;
#endif
        rc = launch_moe_gemm_fp8w_s128(MODE_GATE_UP, q1, max_tiles, s);
        if (rc != SGLK_OK) return rc;
        mark(2);
        A8GemmParams q2{};
        q2.x = ic1q;
        q2.x_stride = 2 * (int64_t)N;
        q2.x_bytes = (int64_t)M * topk * 2 * N;
        q2.xs = ic1s;
        q2.xs_stride = ic1s_stride;
        q2.sorted_slot = sorted_slot;
        q2.topk = topk;
        q2.w = (const uint8_t*)a->w2;
        q2.w_expert_stride = (int64_t)K * N;
        q2.w_scale = a->w2_scale;
        q2.scale_rows = (int)ceil_div(K, a->block_n);
        q2.scale_cols = N / 128;
        q2.block_n = a->block_n;
        q2.C = N;
        q2.tile_info = (const int4*)tile_info;
        q2.num_tiles = num_tiles;
        q2.n_tiles = K / 256;
        q2.out = ic2;
        q2.out_stride = K;
        q2.topk_weights = a->topk_weights;
        q2.max_mtiles = max_tiles;
#ifdef SGLK_DEV_ABLATE
        if (q1.dbg) q2.dbg = q1.dbg + 32 * 16384;
#endif
        rc = launch_moe_gemm_fp8w_s128(MODE_DOWN, q2, max_tiles, s);
        if (rc != SGLK_OK) return rc;
        mark(3);
    } else if (a8) {
        // quantise hidden (one pass), GEMM-1 + SiLU*mul + ic1 quantisation, GEMM-2 + routing weight (moe_gemm_fp8w_s128.hip, one e4m3 term)
        uint8_t* xq = ws + w.xq;
        uint8_t* xs = ws + w.xs;
        uint8_t* ic1q = ws + w.ic1q;
        uint8_t* ic1s = ws + w.ic1s;
        const int xs_stride = (int)align_up(K / 128, 4), ic1s_stride = (int)align_up(N / 128, 4);
        if (!split_done) {
            rc = launch_quant_fp8_block128((const uint16_t*)a->hidden, a->hidden_stride, xq, K, xs, xs_stride, M, K, s);
            if (rc != SGLK_OK) return rc;
            mark(1);   // the quantisation pass counts towards the align stage
        }
        A8GemmParams q1{};
        q1.x = xq;
        q1.x_stride = K;
        q1.x_bytes = (int64_t)M * K;
        q1.xs = xs;
        q1.xs_stride = xs_stride;
        q1.sorted_slot = sorted_slot;
        q1.topk = topk;
        q1.w = (const uint8_t*)a->w1;
        q1.w_expert_stride = (int64_t)2 * N * K;
        q1.w_scale = a->w1_scale;
        q1.scale_rows = (int)ceil_div(2 * N, a->block_n);
        q1.scale_cols = K / 128;
        q1.block_n = a->block_n;
        q1.C = K;
        q1.n_half = N;
        q1.tile_info = (const int4*)tile_info;
        q1.num_tiles = num_tiles;
        q1.n_tiles = N / 128;
        q1.out = ic1q;
        q1.out_stride = N;
        q1.out_s = ic1s;
        q1.out_s_stride = ic1s_stride;
#ifdef SGLK_DEV_ABLATE
        if (knobs().dbg_ptr) q1.dbg = (unsigned long long*)knobs().dbg_ptr;
#endif
        q1.max_mtiles = max_tiles;
        rc = launch_moe_gemm_fp8w_s128(MODE_GATE_UP, q1, max_tiles, s, 1);
        if (rc != SGLK_OK) return rc;
        mark(2);
        A8GemmParams q2{};
        q2.x = ic1q;
        q2.x_stride = N;
        q2.x_bytes = (int64_t)M * topk * N;
        q2.xs = ic1s;
        q2.xs_stride = ic1s_stride;
        q2.sorted_slot = sorted_slot;
        q2.topk = topk;
        q2.w = (const uint8_t*)a->w2;
        q2.w_expert_stride = (int64_t)K * N;
        q2.w_scale = a->w2_scale;
        q2.scale_rows = (int)ceil_div(K, a->block_n);
        q2.scale_cols = N / 128;
        q2.block_n = a->block_n;
        q2.C = N;
        q2.tile_info = (const int4*)tile_info;
        q2.num_tiles = num_tiles;
        q2.n_tiles = K / 256;
        q2.out = ic2;
        q2.out_stride = K;
        q2.topk_weights = a->topk_weights;
#ifdef SGLK_DEV_ABLATE
        if (q1.dbg) q2.dbg = q1.dbg + 32 * 16384;
#endif
        q2.max_mtiles = max_tiles;
        rc = launch_moe_gemm_fp8w_s128(MODE_DOWN, q2, max_tiles, s, 1);
        if (rc != SGLK_OK) return rc;
        mark(3);
    } else if (tuned) {
        MoeGemmParams g1{};
        g1.x = (const uint16_t*)a->hidden;
        g1.x_stride = a->hidden_stride;
        g1.x_bytes = (int64_t)M * a->hidden_stride * 2;
        g1.sorted_slot = sorted_slot;
        g1.topk = topk;
        g1.w = (const uint8_t*)a->w1;
        g1.w_expert_stride = (int64_t)2 * N * K;
        g1.w_scale = a->w1_scale;
        g1.scale_rows = (int)ceil_div(2 * N, a->block_n);
        g1.scale_cols = K / 128;
        g1.block_n = a->block_n;
        g1.C = K;
        g1.n_half = N;
        g1.tile_info = (const int4*)tile_info;
        g1.num_tiles = num_tiles;
        g1.n_tiles = tile_m == 128 ? N / 64 : N / 128;
        g1.out = ic1;
        g1.out_stride = N;
        g1.topk_weights = nullptr;
        if (inline_align) { g1.inline_ids = a->topk_ids; g1.inline_slots = M * topk; g1.inline_experts = E; }
        // decode-size kernels (stream / mid): non-temporal weight reads from 32 routed rows on (ld_stream16; SGLK_W_NT=0/1 forces)
        const int w_nt = knobs().w_nt >= 0 ? knobs().w_nt : ((int64_t)M * topk >= 32 ? 1 : 0);
        g1.w_nt = w_nt;
        // per-XCD tile tickets of the two persistent launches (8 counters each), zeroed per call by moe_align's last launch
        if (tile_m == 256) g1.tickets = (int*)(ws + w.tickets);
#ifdef SGLK_DEV_ABLATE
        if (knobs().dbg_ptr) g1.dbg = (unsigned long long*)knobs().dbg_ptr;
#endif
        hipEvent_t ev_join = nullptr;
        bool tails_last = false;     // SGLK_TAIL_SPLIT=2: the tails on the caller's stream, BEHIND the two big launches
        MoeGemmParams tl1{}, tl2{};
        const int tails_max = E < max_tiles ? E : max_tiles;
        if (split_tails) {
            MoeGemmParams t1 = g1;
            t1.tile_info = (const int4*)tile_info_b;
            t1.num_tiles = num_tiles_b;
            t1.tickets = nullptr;
            t1.w_nt = 0;              // the big launch reads the same experts: leave their weights cacheable
            MoeGemmParams t2{};
            t2.x = ic1;
            t2.x_stride = N;
            t2.x_bytes = (int64_t)M * topk * N * 2;
            t2.sorted_slot = sorted_slot;
            t2.topk = topk;
            t2.w = (const uint8_t*)a->w2;
            t2.w_expert_stride = (int64_t)K * N;
            t2.w_scale = a->w2_scale;
            t2.scale_rows = (int)ceil_div(K, a->block_n);
            t2.scale_cols = N / 128;
            t2.block_n = a->block_n;
            t2.C = N;
            t2.tile_info = (const int4*)tile_info_b;
            t2.num_tiles = num_tiles_b;
            t2.n_tiles = K / 128;
            t2.out = ic2;
            t2.out_stride = K;
            t2.topk_weights = a->topk_weights;
            hipStream_t ts = s;
            if (side) {   // fork: the side stream starts once moe_align (and the ticket reset) are done
                hipEvent_t ev_fork = (hipEvent_t)a->aux_events[0];
                ev_join = (hipEvent_t)a->aux_events[1];
                if (hipEventRecord(ev_fork, s) != hipSuccess || hipStreamWaitEvent((hipStream_t)a->aux_stream, ev_fork, 0) != hipSuccess)
                    SGLK_FAIL(SGLK_ERR_LAUNCH, "fused_experts: aux-stream fork failed");
                ts = (hipStream_t)a->aux_stream;
            }
            // with a side stream both tail launches go out here, before the big GEMM-1, and overlap it and GEMM-2; on the
            // caller's stream they simply run first
            tails_last = !side && tail_knob == 2;
            if (tails_last) {
                tl1 = t1;
                tl2 = t2;
            } else {
                rc = launch_moe_gemm_fp8w_mid(MODE_GATE_UP, t1, tails_max, ts);
                if (rc != SGLK_OK) return rc;
                rc = launch_moe_gemm_fp8w_mid(MODE_DOWN, t2, tails_max, ts);
                if (rc != SGLK_OK) return rc;
            }
            if (side && hipEventRecord(ev_join, ts) != hipSuccess) SGLK_FAIL(SGLK_ERR_LAUNCH, "fused_experts: aux-stream record failed");
        }
        rc = tile_m == 256 ? launch_moe_gemm_fp8w_256i(MODE_GATE_UP, g1, max_tiles, s)
             : tile_m == kStreamTileM ? launch_moe_gemm_fp8w_stream(MODE_GATE_UP, g1, max_tiles, s)
             : tile_m == kMidTileM    ? launch_moe_gemm_fp8w_mid(MODE_GATE_UP, g1, max_tiles, s)
                                      : launch_moe_gemm_fp8w(MODE_GATE_UP, g1, max_tiles, s);
        if (rc != SGLK_OK) return rc;
        mark(2);

        MoeGemmParams g2{};
        g2.x = ic1;
        g2.x_stride = N;
        g2.x_bytes = (int64_t)M * topk * N * 2;
        g2.sorted_slot = sorted_slot;
        g2.topk = topk;
        g2.w = (const uint8_t*)a->w2;
        g2.w_expert_stride = (int64_t)K * N;
        g2.w_scale = a->w2_scale;
        g2.scale_rows = (int)ceil_div(K, a->block_n);
        g2.scale_cols = N / 128;
        g2.block_n = a->block_n;
        g2.C = N;
        g2.n_half = 0;
        g2.w_nt = w_nt;
        g2.tile_info = (const int4*)tile_info;
        g2.num_tiles = num_tiles;
        if (inline_align) { g2.inline_ids = a->topk_ids; g2.inline_slots = M * topk; g2.inline_experts = E; }
        // mid kernel, DOWN: two 128-column tiles per workgroup on one weight ring when the shapes allow (SGLK_MID_DOWN2=0: one)
        const bool mid_down2 = tile_m == kMidTileM && N % 256 == 0 && K % 256 == 0 && knobs().mid_down2 != 0;
        g2.n_tiles = (tile_m == 128 || (tile_m == kMidTileM && !mid_down2)) ? K / 128 : K / 256;
        g2.out = ic2;
        g2.out_stride = K;
        g2.topk_weights = a->topk_weights;
        if (g1.tickets) g2.tickets = g1.tickets + 8;
#ifdef SGLK_DEV_ABLATE
        if (g1.dbg) g2.dbg = g1.dbg + 32 * 16384;
#endif
        rc = tile_m == 256 ? launch_moe_gemm_fp8w_256i(MODE_DOWN, g2, max_tiles, s)
             : tile_m == kStreamTileM ? launch_moe_gemm_fp8w_stream(MODE_DOWN, g2, max_tiles, s)
             : tile_m == kMidTileM    ? (mid_down2 ? launch_moe_gemm_fp8w_mid_down2(g2, max_tiles, s)
                                                    : launch_moe_gemm_fp8w_mid(MODE_DOWN, g2, max_tiles, s))
                                      : launch_moe_gemm_fp8w(MODE_DOWN, g2, max_tiles, s);
        if (rc != SGLK_OK) return rc;
        if (tails_last) {
            rc = launch_moe_gemm_fp8w_mid(MODE_GATE_UP, tl1, tails_max, s);
            if (rc != SGLK_OK) return rc;
            rc = launch_moe_gemm_fp8w_mid(MODE_DOWN, tl2, tails_max, s);
            if (rc != SGLK_OK) return rc;
        }
        // join: the combine needs the tail tiles' rows of ic2 too
        if (ev_join && hipStreamWaitEvent(s, ev_join, 0) != hipSuccess) SGLK_FAIL(SGLK_ERR_LAUNCH, "fused_experts: side-stream join failed");
        mark(3);
    } else if (mid_i8) {
        // the same four steps as the 256-row int8 path below, with the weight-streaming kernel for the two GEMMs
        int8_t* xq = (int8_t*)(ws + w.xq);
        float* xs = (float*)(ws + w.xs);
        rc = launch_quant_int8_rows((const uint16_t*)a->hidden, a->hidden_stride, xq, K, xs, M, K, 1e-7f, s);
        if (rc != SGLK_OK) return rc;
        I8GemmParams q1{};
        q1.x = xq;
        q1.x_stride = K;
        q1.x_scale = xs;
        q1.w = (const uint8_t*)a->w1;
        q1.w_bytes = (int64_t)2 * N * K;
        q1.w_scale = a->w1_scale;
        q1.scale_rows = 2 * N;
        q1.out = ic1;                  // fp32 [position][N]
        q1.out_stride = N;
        q1.M = M;
        q1.K = K;
        q1.n_tiles = N / 128;
        q1.tile_info = (const int4*)tile_info;
        q1.num_tiles = num_tiles;
        q1.sorted_slot = sorted_slot;
        q1.topk = topk;
        q1.n_half = N;
        q1.w_nt = knobs().w_nt >= 0 ? knobs().w_nt : ((int64_t)M * topk >= 32 ? 1 : 0);   // as on the fp8 path
        rc = launch_gemm_i8_mid(MODE_GATE_UP, q1, max_tiles, s);
        if (rc != SGLK_OK) return rc;
        mark(2);
        int8_t* hq = (int8_t*)(ws + w.ic1q);
        float* hs = (float*)(ws + w.ic1s);
        rc = launch_quant_int8_rows_f32((const float*)ic1, N, hq, N, hs, (int64_t)M * topk, N, 1e-7f, s);
        if (rc != SGLK_OK) return rc;
        I8GemmParams q2{};
        q2.x = hq;
        q2.x_stride = N;
        q2.x_scale = hs;
        q2.w = (const uint8_t*)a->w2;
        q2.w_bytes = (int64_t)K * N;
        q2.w_scale = a->w2_scale;
        q2.scale_rows = K;
        q2.out = ic2;                  // bf16 [slot][K]
        q2.out_stride = K;
        q2.M = M * topk;
        q2.K = N;
        q2.n_tiles = K / 128;
        q2.tile_info = (const int4*)tile_info;
        q2.num_tiles = num_tiles;
        q2.sorted_slot = sorted_slot;
        q2.topk = topk;
        q2.topk_weights = a->topk_weights;
        q2.w_nt = q1.w_nt;
        rc = launch_gemm_i8_mid(MODE_DOWN, q2, max_tiles, s);
        if (rc != SGLK_OK) return rc;
        mark(3);
    } else if (i8s) {
        // W8A8 on the 128-token kernel (moe_gemm_fp8w_s128.hip, terms = 0): x per token -> int8 (inside the align stage), GEMM-1
        // + SiLU*mul + the per-token quantisation of ic1 in ONE launch (row maxima exchanged between the m-tile's workgroups),
        // GEMM-2 + routing weight; same arithmetic, bit for bit, as the 256-row path below (/root/reference/test_moe_int8.py:59-94)
        int8_t* xq = (int8_t*)(ws + w.xq);
        float* xs = (float*)(ws + w.xs);
        if (!split_done) {
            rc = launch_quant_int8_rows((const uint16_t*)a->hidden, a->hidden_stride, xq, K, xs, M, K, 1e-7f, s);
            if (rc != SGLK_OK) return rc;
        }
        if (hipMemsetAsync(ws + w.i8_sync, 0, w.i8_sync_bytes, s) != hipSuccess) SGLK_FAIL(SGLK_ERR_LAUNCH, "fused_experts: memset failed");
        mark(1);   // both count towards the align stage
        int8_t* hq = (int8_t*)(ws + w.ic1q);
        float* hs = (float*)(ws + w.ic1s);
        A8GemmParams q1{};
        q1.x = (const uint8_t*)xq;
        q1.x_stride = K;
        q1.x_bytes = (int64_t)M * K;
        q1.x_scale_f32 = xs;
        q1.sorted_slot = sorted_slot;
        q1.topk = topk;
        q1.w = (const uint8_t*)a->w1;
        q1.w_expert_stride = (int64_t)2 * N * K;
        q1.w_scale = a->w1_scale;
        q1.scale_rows = 2 * N;
        q1.C = K;
        q1.n_half = N;
        q1.tile_info = (const int4*)tile_info;
        q1.num_tiles = num_tiles;
        q1.n_tiles = N / 128;
        q1.out = hq;
        q1.out_stride = N;
        q1.out_scale_f32 = hs;
        q1.row_amax = (unsigned*)(ws + w.i8_sync);
        q1.arrivals = (int*)(ws + w.i8_sync + (size_t)M * topk * sizeof(unsigned));
        q1.quant_floor = 1e-7f;
        q1.max_mtiles = max_tiles;
#ifdef SGLK_DEV_ABLATE
        if (knobs().dbg_ptr) q1.dbg = (unsigned long long*)knobs().dbg_ptr;
#endif
        rc = launch_moe_gemm_fp8w_s128(MODE_GATE_UP, q1, max_tiles, s, 0);
        if (rc != SGLK_OK) return rc;
        mark(2);
        A8GemmParams q2{};
        q2.x = (const uint8_t*)hq;
        q2.x_stride = N;
        q2.x_bytes = (int64_t)M * topk * N;
        q2.x_scale_f32 = hs;
        q2.sorted_slot = sorted_slot;
        q2.topk = topk;
        q2.w = (const uint8_t*)a->w2;
        q2.w_expert_stride = (int64_t)K * N;
        q2.w_scale = a->w2_scale;
        q2.scale_rows = K;
        q2.C = N;
        q2.tile_info = (const int4*)tile_info;
        q2.num_tiles = num_tiles;
        q2.n_tiles = K / 256;
        q2.out = ic2;
        q2.out_stride = K;
        q2.topk_weights = a->topk_weights;
        q2.max_mtiles = max_tiles;
#ifdef SGLK_DEV_ABLATE
        if (q1.dbg) q2.dbg = q1.dbg + 32 * 16384;
#endif
        rc = launch_moe_gemm_fp8w_s128(MODE_DOWN, q2, max_tiles, s, 0);
        if (rc != SGLK_OK) return rc;
        mark(3);
    } else if (tuned_i8) {
        // W8A8 on the int8 matrix cores: quantise x per token, GEMM-1 (+SiLU*mul, fp32 out), quantise ic1 per row,
        // GEMM-2 (+routing weight, bf16 rows by slot); /root/reference/test_moe_int8.py:59-94
        int8_t* xq = (int8_t*)(ws + w.xq);
        float* xs = (float*)(ws + w.xs);
        rc = launch_quant_int8_rows((const uint16_t*)a->hidden, a->hidden_stride, xq, K, xs, M, K, 1e-7f, s);
        if (rc != SGLK_OK) return rc;
        I8GemmParams q1{};
        q1.x = xq;
        q1.x_stride = K;
        q1.x_bytes = (int64_t)M * K;
        q1.x_scale = xs;
        q1.w = (const uint8_t*)a->w1;
        q1.w_bytes = (int64_t)2 * N * K;
        q1.w_scale = a->w1_scale;
        q1.scale_rows = 2 * N;
        q1.out = ic1;                  // fp32 [position][N]
        q1.out_stride = N;
        q1.M = M;
        q1.K = K;
        q1.n_tiles = N / 128;
        q1.tile_info = (const int4*)tile_info;
        q1.num_tiles = num_tiles;
        q1.sorted_slot = sorted_slot;
        q1.topk = topk;
        q1.n_half = N;
        rc = launch_gemm_i8_256(MODE_GATE_UP, q1, max_tiles, s);
        if (rc != SGLK_OK) return rc;
        mark(2);
        int8_t* hq = (int8_t*)(ws + w.ic1q);
        float* hs = (float*)(ws + w.ic1s);
        rc = launch_quant_int8_rows_f32((const float*)ic1, N, hq, N, hs, (int64_t)M * topk, N, 1e-7f, s);
        if (rc != SGLK_OK) return rc;
        I8GemmParams q2{};
        q2.x = hq;
        q2.x_stride = N;
        q2.x_bytes = (int64_t)M * topk * N;
        q2.x_scale = hs;
        q2.w = (const uint8_t*)a->w2;
        q2.w_bytes = (int64_t)K * N;
        q2.w_scale = a->w2_scale;
        q2.scale_rows = K;
        q2.out = ic2;                  // bf16 [slot][K]
        q2.out_stride = K;
        q2.M = M * topk;
        q2.K = N;
        q2.n_tiles = K / 256;
        q2.tile_info = (const int4*)tile_info;
        q2.num_tiles = num_tiles;
        q2.sorted_slot = sorted_slot;
        q2.topk = topk;
        q2.topk_weights = a->topk_weights;
        rc = launch_gemm_i8_256(MODE_DOWN, q2, max_tiles, s);
        if (rc != SGLK_OK) return rc;
        mark(3);
    } else if (mid_b16) {
        BmidParams q1{};
        q1.x = (const uint16_t*)a->hidden;
        q1.x_stride = a->hidden_stride;
        q1.w = (const uint8_t*)a->w1;
        q1.w_expert_stride = (int64_t)2 * N * K * 2;
        q1.out = ic1;                  // bf16 [position][N]
        q1.out_stride = N;
        q1.M = M; q1.N = N; q1.K = K;
        q1.tile_info = (const int4*)tile_info;
        q1.num_tiles = num_tiles;
        q1.sorted_slot = sorted_slot;
        q1.topk = topk;
        q1.n_half = N;
        rc = launch_moe_gemm_bf16_mid(MODE_GATE_UP, q1, max_tiles, s);
        if (rc != SGLK_OK) return rc;
        mark(2);
        BmidParams q2{};
        q2.x = ic1;
        q2.x_stride = N;
        q2.w = (const uint8_t*)a->w2;
        q2.w_expert_stride = (int64_t)K * N * 2;
        q2.out = ic2;                  // bf16 [slot][K]
        q2.out_stride = K;
        q2.M = M * topk; q2.N = K; q2.K = N;
        q2.tile_info = (const int4*)tile_info;
        q2.num_tiles = num_tiles;
        q2.sorted_slot = sorted_slot;
        q2.topk = topk;
        q2.topk_weights = a->topk_weights;
        rc = launch_moe_gemm_bf16_mid(MODE_DOWN, q2, max_tiles, s);
        if (rc != SGLK_OK) return rc;
        mark(3);
    } else if (tuned_b16) {
        // bf16 experts on the bf16 matrix cores, weights in the reference's packed order (/root/reference/test_moe.py:79-92)
        Bf16GemmParams q1{};
        q1.x = (const uint16_t*)a->hidden;
        q1.x_stride = a->hidden_stride * 2;
        q1.x_bytes = (int64_t)M * a->hidden_stride * 2;
        q1.w = (const uint8_t*)a->w1;
        q1.w_bytes = (int64_t)2 * N * K * 2;
        q1.out = ic1;                  // bf16 [position][N]
        q1.out_stride = N;
        q1.M = M;
        q1.K = K;
        q1.n_tiles = N / 128;
        q1.tile_info = (const int4*)tile_info;
        q1.num_tiles = num_tiles;
        q1.sorted_slot = sorted_slot;
        q1.topk = topk;
        q1.n_half = N;
        rc = launch_gemm_bf16_256(MODE_GATE_UP, q1, max_tiles, s);
        if (rc != SGLK_OK) return rc;
        mark(2);
        Bf16GemmParams q2{};
        q2.x = ic1;
        q2.x_stride = (int64_t)N * 2;
        q2.x_bytes = (int64_t)M * topk * N * 2;
        q2.w = (const uint8_t*)a->w2;
        q2.w_bytes = (int64_t)K * N * 2;
        q2.out = ic2;                  // bf16 [slot][K]
        q2.out_stride = K;
        q2.M = M * topk;
        q2.K = N;
        q2.n_tiles = K / 256;
        q2.tile_info = (const int4*)tile_info;
        q2.num_tiles = num_tiles;
        q2.sorted_slot = sorted_slot;
        q2.topk = topk;
        q2.topk_weights = a->topk_weights;
        rc = launch_gemm_bf16_256(MODE_DOWN, q2, max_tiles, s);
        if (rc != SGLK_OK) return rc;
        mark(3);
    } else {
        // generic engine: any shape / weight type / row-major weights (gemm_generic.hip)
        const int wes = a->wtype == SGLK_W_BF16 ? 2 : 1;
        GenericGemmParams g1{};
        g1.x = a->hidden;
        g1.x_type = SGLK_W_BF16;
        g1.x_stride = a->hidden_stride;
        if (a->wtype == SGLK_W_INT8) {
            int8_t* xq = (int8_t*)(ws + w.xq);
            float* xs = (float*)(ws + w.xs);
            rc = launch_quant_int8_rows((const uint16_t*)a->hidden, a->hidden_stride, xq, K, xs, M, K, 1e-7f, s);
            if (rc != SGLK_OK) return rc;
            g1.x = xq;
            g1.x_type = SGLK_W_INT8;
            g1.x_stride = K;
            g1.x_row_scale = xs;
        }
        g1.sorted_slot = sorted_slot;
        g1.topk = topk;
        g1.gather = GG_GATHER_TOKEN;
        g1.tile_info = (const int4*)tile_info;
        g1.num_tiles = num_tiles;
        g1.n_tiles = (int)ceil_div(N, 32);
        g1.w = a->w1;
        g1.w_type = a->wtype;
        g1.packed = a->packed & 1;
        g1.w_expert_stride = (int64_t)2 * N * K * wes;
        g1.C = K;
        g1.w_scale = a->w1_scale;
        if (a->wtype == SGLK_W_FP8_E4M3) {
            g1.scale_rows = (int)ceil_div(2 * N, a->block_n);
            g1.scale_cols = (int)ceil_div(K, 128);
            g1.block_n = a->block_n;
        } else {
            g1.scale_rows = 2 * N;   // int8: one scale per weight row
            g1.scale_cols = 1;
            g1.block_n = 1;
        }
        g1.n_half = N;
        g1.n_out = N;
        g1.out = ic1;
        g1.out_type = a->wtype == SGLK_W_INT8 ? SGLK_OUT_F32 : SGLK_OUT_BF16;
        g1.out_stride = N;
        rc = launch_gemm_generic(GG_GATE_UP, g1, max_tiles, s);
        if (rc != SGLK_OK) return rc;
        mark(2);

        GenericGemmParams g2{};
        g2.x = ic1;
        g2.x_type = SGLK_W_BF16;
        g2.x_stride = N;
        if (a->wtype == SGLK_W_INT8) {
            int8_t* q = (int8_t*)(ws + w.ic1q);
            float* qs = (float*)(ws + w.ic1s);
            // rows of ic1 are positions; only the first (#valid slots) are defined, the rest is never read back
            rc = launch_quant_int8_rows_f32((const float*)ic1, N, q, N, qs, (int64_t)M * topk, N, 1e-7f, s);
            if (rc != SGLK_OK) return rc;
            g2.x = q;
            g2.x_type = SGLK_W_INT8;
            g2.x_row_scale = qs;
        }
        g2.sorted_slot = sorted_slot;
        g2.topk = topk;
        g2.gather = GG_GATHER_NONE;
        g2.tile_info = (const int4*)tile_info;
        g2.num_tiles = num_tiles;
        g2.n_tiles = (int)ceil_div(K, 64);
        g2.w = a->w2;
        g2.w_type = a->wtype;
        g2.packed = (a->packed >> 1) & 1;
        g2.w_expert_stride = (int64_t)K * N * wes;
        g2.C = N;
        g2.w_scale = a->w2_scale;
        if (a->wtype == SGLK_W_FP8_E4M3) {
            g2.scale_rows = (int)ceil_div(K, a->block_n);
            g2.scale_cols = (int)ceil_div(N, 128);
            g2.block_n = a->block_n;
        } else {
            g2.scale_rows = K;
            g2.scale_cols = 1;
            g2.block_n = 1;
        }
        g2.n_out = K;
        g2.out = ic2;
        g2.out_type = SGLK_OUT_BF16;
        g2.out_stride = K;
        g2.scatter = 1;
        g2.topk_weights = a->topk_weights;
        rc = launch_gemm_generic(GG_DOWN, g2, max_tiles, s);
        if (rc != SGLK_OK) return rc;
        mark(3);
    }

    if (shared) {
        // moe block: the shared expert's last launch sums the routed slots itself (x routed_scaling_factor) -- no combine
        // launch, no [M][K] round trip of the routed output
        const MoeSlotAddend moe{ic2, a->topk_ids, topk, E};
        rc = shared_expert_impl(shared, stream, &moe);
    } else {
        rc = launch_moe_combine(ic2, a->topk_ids, (uint16_t*)a->out, a->out_stride, M, K, E, topk, s);
    }
    mark(4);
    if (a->path_taken) {
        int path = (tile_m & SGLK_PATH_TILE_MASK) | (a8 ? SGLK_PATH_FP8_ACT : 0) | (split ? SGLK_PATH_SPLIT : 0);
        if (split_tails) path |= SGLK_PATH_TAILS_SPLIT | (side ? SGLK_PATH_TAILS_AUX : 0);
        if (tuned && !a8 && !split && tile_m == 256) {
            if (moe_gemm_fp8w_256i_is_persistent(K, (int64_t)max_tiles * (N / 128))) path |= SGLK_PATH_PERSIST_G1;
            if (moe_gemm_fp8w_256i_is_persistent(N, (int64_t)max_tiles * (K / 256))) path |= SGLK_PATH_PERSIST_G2;
        }
        *a->path_taken = path | (inline_align ? SGLK_PATH_INLINE_ALIGN : 0) | (routed_and_aligned ? SGLK_PATH_ROUTE_ALIGN : 0) | (shared ? SGLK_PATH_SHARED_FOLDED : 0);
    }
    return rc;
}
}  // namespace

extern "C" size_t sglk_moe_block_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t E, int32_t topk, int32_t wtype,
                                                 int32_t flags, int32_t shared_N) {
    if (M < 0 || N <= 0 || K <= 0 || E <= 0 || topk <= 0) return 0;
    size_t t = align_up(plan_workspace(M, N, K, E, topk, wtype, flags).total, 256);
    if (shared_N > 0) t += align_up(sglk_shared_expert_workspace_bytes(M, shared_N, K, wtype), 256) + align_up((size_t)M * K * 2, 256);
    return t;
}

extern "C" int sglk_moe_block(const sglk_moe_block_args* b, void* stream) {
    SGLK_REQUIRE(b, SGLK_ERR_INVALID, "moe_block: null args");
    const sglk_fused_experts_args& ex0 = b->experts;
    SGLK_REQUIRE(b->gating && (ex0.M == 0 || (ex0.topk_weights && ex0.topk_ids)), SGLK_ERR_INVALID, "moe_block: null pointer");
    SGLK_REQUIRE(b->gating_type >= 0 && b->gating_type <= 2 && b->gating_stride >= ex0.E, SGLK_ERR_INVALID, "moe_block: gating");
    const int M = ex0.M, N = ex0.N, K = ex0.K, E = ex0.E, topk = ex0.topk;
    SGLK_REQUIRE(M >= 0 && N > 0 && K > 0 && E > 0 && topk > 0, SGLK_ERR_INVALID, "moe_block: bad sizes");
    const size_t ws_ex = align_up(plan_workspace(M, N, K, E, topk, ex0.wtype, ex0.flags).total, 256);
    const size_t ws_sh = b->shared_N > 0 ? align_up(sglk_shared_expert_workspace_bytes(M, b->shared_N, K, ex0.wtype), 256) : 0;
    const size_t ws_tmp = b->shared_N > 0 ? align_up((size_t)M * K * 2, 256) : 0;
    SGLK_REQUIRE(ex0.workspace && ex0.workspace_bytes >= ws_ex + ws_sh + ws_tmp, SGLK_ERR_WORKSPACE,
                 "moe_block: workspace %zu < required %zu", ex0.workspace_bytes, ws_ex + ws_sh + ws_tmp);
    const RouteArgs route{b->gating, b->gating_stride, b->gating_type, b->correction_bias, b->renormalize, b->num_expert_group,
                          b->topk_group};
    sglk_fused_experts_args ex = ex0;
    ex.workspace_bytes = ws_ex;
    if (b->shared_N <= 0) return fused_experts_impl(&ex, stream, &route, nullptr);
    SGLK_REQUIRE(b->shared_w1 && b->shared_w2, SGLK_ERR_INVALID, "moe_block: shared expert weights missing");
    unsigned char* wsb = (unsigned char*)ex0.workspace;
    sglk_shared_expert_args sh{};
    sh.hidden = ex0.hidden;
    sh.hidden_stride = ex0.hidden_stride;
    sh.out = ex0.out;
    sh.out_stride = ex0.out_stride;
    sh.w1 = b->shared_w1;
    sh.w2 = b->shared_w2;
    sh.w1_scale = b->shared_w1_scale;
    sh.w2_scale = b->shared_w2_scale;
    sh.routed_scaling_factor = b->routed_scaling_factor;
    sh.M = M;
    sh.N = b->shared_N;
    sh.K = K;
    sh.wtype = ex0.wtype;
    sh.packed = b->shared_packed;
    sh.block_n = ex0.block_n;
    sh.block_k = ex0.block_k;
    sh.workspace = wsb + ws_ex;
    sh.workspace_bytes = ws_sh;
    if (M > 0 && shared_expert_can_fold(&sh) && !knobs().no_block_fold) return fused_experts_impl(&ex, stream, &route, &sh);
    // the shared expert's path for this shape has no slot addend: routed experts into a scratch [M][K], then the plain call
    uint16_t* tmp = (uint16_t*)(wsb + ws_ex + ws_sh);
    ex.out = tmp;
    ex.out_stride = K;
    int rc = fused_experts_impl(&ex, stream, &route, nullptr);
    if (rc != SGLK_OK) return rc;
    sh.fused_out = tmp;
    sh.fused_out_stride = K;
    return shared_expert_impl(&sh, stream, nullptr);
}

extern "C" void* sglk_stage_timer_create(int32_t max_calls) {
    if (max_calls <= 0) return nullptr;
    StageTimer* t = new StageTimer();
    t->max_calls = max_calls;
    t->ev.resize((size_t)max_calls * (SGLK_NUM_STAGES + 1));
    for (auto& e : t->ev) {
        if (hipEventCreate(&e) != hipSuccess) {
            set_error("stage_timer: hipEventCreate failed");
            delete t;
            return nullptr;
        }
    }
    return t;
}

extern "C" void sglk_stage_timer_destroy(void* timer) {
    StageTimer* t = (StageTimer*)timer;
    if (!t) return;
    for (auto& e : t->ev) hipEventDestroy(e);
    delete t;
}

extern "C" void sglk_stage_timer_reset(void* timer) {
    if (timer) ((StageTimer*)timer)->calls = 0;
}

extern "C" int sglk_stage_timer_read(void* timer, float* mean_ms, int32_t* calls) {
    StageTimer* t = (StageTimer*)timer;
    SGLK_REQUIRE(t && mean_ms && calls, SGLK_ERR_INVALID, "stage_timer_read: null pointer");
    *calls = t->calls;
    for (int i = 0; i < SGLK_NUM_STAGES; ++i) mean_ms[i] = 0.f;
    if (t->calls == 0) return SGLK_OK;
    if (hipEventSynchronize(t->at(t->calls - 1, SGLK_NUM_STAGES)) != hipSuccess)
        SGLK_FAIL(SGLK_ERR_LAUNCH, "stage_timer_read: hipEventSynchronize failed");
    for (int c = 0; c < t->calls; ++c)
        for (int i = 0; i < SGLK_NUM_STAGES; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, t->at(c, i), t->at(c, i + 1)) != hipSuccess)
                SGLK_FAIL(SGLK_ERR_LAUNCH, "stage_timer_read: hipEventElapsedTime failed");
            mean_ms[i] += ms;
        }
    for (int i = 0; i < SGLK_NUM_STAGES; ++i) mean_ms[i] /= (float)t->calls;
    return SGLK_OK;
}

extern "C" int sglk_quant_fp8_block128(const void* x, int64_t x_stride, void* q, int64_t q_stride, void* scale,
                                       int64_t scale_stride, int64_t rows, int32_t cols, void* stream) {
    SGLK_REQUIRE(rows >= 0 && cols > 0, SGLK_ERR_INVALID, "quant_fp8_block128: bad sizes");
    SGLK_REQUIRE(rows == 0 || (x && q && scale), SGLK_ERR_INVALID, "quant_fp8_block128: null pointer");
    SGLK_REQUIRE(x_stride >= cols && q_stride >= cols && scale_stride >= cols / 128, SGLK_ERR_INVALID, "quant_fp8_block128: stride");
    return launch_quant_fp8_block128((const uint16_t*)x, x_stride, (uint8_t*)q, q_stride, (uint8_t*)scale, scale_stride, rows, cols,
                                     (hipStream_t)stream);
}

extern "C" int sglk_split_fp8_block128(const void* x, int64_t x_stride, void* q, int64_t q_stride, void* scale,
                                       int64_t scale_stride, int64_t rows, int32_t cols, void* stream) {
    SGLK_REQUIRE(rows >= 0 && cols > 0, SGLK_ERR_INVALID, "split_fp8_block128: bad sizes");
    SGLK_REQUIRE(rows == 0 || (x && q && scale), SGLK_ERR_INVALID, "split_fp8_block128: null pointer");
    SGLK_REQUIRE(x_stride >= cols && q_stride >= 2 * (int64_t)cols && scale_stride >= cols / 128, SGLK_ERR_INVALID, "split_fp8_block128: stride");
    return launch_split_fp8_block128((const uint16_t*)x, x_stride, (uint8_t*)q, q_stride, (uint8_t*)scale, scale_stride, rows, cols,
                                     (hipStream_t)stream);
}
