// Dense entry points of the C-ABI built on the generic engine: shared_expert, scaled_mm (fp8 / int8 / bf16 weights).
#include <stdlib.h>

#include "knobs.h"
#include "moe_internal.h"

using namespace sglk;

namespace {

struct DenseWs {
    size_t tile_info, num_tiles, ident, ic1, xq, xs, ic1q, ic1s, partial, xsplit, xsplit_s, total;
};

// dense GEMMs on the 128-token two-workgroups-per-CU kernel (moe_gemm_fp8w_s128.hip, MODE_PLAIN): whole 256-column output tiles,
// the reduction in whole 128-wide blocks (2 .. 64), and at least two full rounds of its 128 x 256 tiles over the chip's 512
// workgroup slots -- below that the 256-row kernels' larger tiles win (same-box A/B, profiles/r03_ab_dense_s128.txt:
// 4096 x 12288 x 2048 fp8 999 -> 1149 TFLOP/s, int8 1731 -> 1970 TOP/s; 4096 x 1536 x 2048 and 4096 x 2048 x 6144 lose 7-33 %)
bool dense_s128_shape_ok(int M, int N, int K) {
    const int k = knobs().dense_s128;
    const int64_t wgs = ceil_div(M, 128) * (int64_t)(N / 256);
    return k != 0 && (k > 0 ? M >= 128 : wgs >= 1024) && N % 256 == 0 && K % 128 == 0 && K >= 256 && K <= 8192 && !knobs().force_generic;
}

// the tuned 256-token fp8 kernel (moe_gemm_fp8w_256i.hip) takes a dense [M][C] x [R][C]^T when this holds
bool tuned_dense_ok(int M, int R, int C, int wtype, int packed, int block_n, int block_k, const void* x, int64_t x_stride) {
    return wtype == SGLK_W_FP8_E4M3 && packed && M >= 192 && R % 256 == 0 && C % 128 == 0 && C >= 256 && (C >> 7) <= 64 &&
           block_k == 128 && block_n % 32 == 0 && x_stride % 8 == 0 && ((uintptr_t)x % 16) == 0 &&
           (int64_t)M * x_stride * 2 < (1ll << 32) && (int64_t)R * C < (1ll << 32) && !knobs().force_generic;
}

// int8 shared expert on the weight-streaming kernel: one 128-row tile always, several (M < SGLK_SHARED_I8_MID_MAX) while that beats the
// 256-row kernels' one-tile time (2048 x 7168: 129 ... 191 rows ran the generic engine, 251 us; 192 ... 1024 rows ~117 us whatever M)
// `width` = the expert's intermediate size (gate_up has 2 x width rows)
static bool shared_rows_take_mid(int M, int width, int max_rows) {
    if (M < max_rows) return true;
    // from 1024 rows on: while the tile kernels' gate_up launch would have at most 64 workgroups (width 768: 74 -> 43-55 us at 1024 ...
    // 2000 rows; width 2048 at 1024 rows 189 -> 149 us fp8, and the tile kernels from 1280 rows; profiles/r03_ab_rows_1024_2047.txt)
    return max_rows >= 1024 && knobs().shared_big_wgs > 0 && M < kMidDenseMaxM &&
           ceil_div(M, 256) * (int64_t)(2 * width / 256) <= knobs().shared_big_wgs;
}
static int shared_i8_ksplit(int M, int N, int K, int width) {
    if (M <= 128) return i8_mid_ksplit(M, N, K);
    return shared_rows_take_mid(M, width, knobs().shared_i8_mid_max) ? i8_mid_dense_ksplit(M, N, K) : 0;
}

// Dense GEMM, packed weights: weight-streaming (split-K) kernel or 256-row tile kernel?  Below 192 rows always the former.  From
// there to SGLK_DENSE_MID_MAX (1024) the 256-row kernel only has ceil(M / 256) x N / 256 workgroups -- 16 for a 4096-wide layer --
// and loses to the streaming kernels by 2-4x until it can fill a good part of the chip (same-box A/B, tools/ab_dense_mid.py,
// profiles/r02_ab_dense_mid.txt: 192 x 4096 x 4096 fp8 0.096 -> 0.025 ms, bf16 0.072 -> 0.036; 1000 x 2048 x 6144 fp8 0.118 -> 0.053;
// a 12288-wide bf16 layer stays on the 256-row kernel).  Without split-K that kernel's time is one tile's time (K / 64 stages)
// however few tiles there are.  Round 3, after the split-K of the streaming kernels went to the rounds model (knobs.h: splitk_by_rounds):
// crossover re-measured (tools/ab_dense_crossover.py, profiles/r03_ab_dense_crossover.txt): bf16 up to 60 workgroups of the tile kernel
// (384 x 5120 x 2048: 41 -> 26 us; 192-256 x 12288 x 2048: 41 -> 28), fp8 up to 96 as before, int8 up to 64 (gemm_api.hip below).
bool dense_prefers_mid(int M, int N, int wtype) {
    if (M < 192) return true;
    if (M >= knobs().dense_mid_max || M >= kMidDenseMaxM) return false;
    const int64_t wgs = ceil_div(M, 256) * (int64_t)(N / 256);
    // from 1024 rows on the bf16 tile kernel splits K itself and wins from 40 workgroups on (1280 x 2048 x 6144: 58 vs 69 us); fp8 keeps
    // its threshold up to 2047 rows (1280 x 4096 x 4096: 68 -> 55 us; profiles/r03_ab_rows_1024_2047.txt)
    const int bf16_max = M >= 1024 && knobs().dense_mid_wgs_bf16 > 32 ? 32 : knobs().dense_mid_wgs_bf16;
    return wgs <= (wtype == SGLK_W_FP8_E4M3 ? knobs().dense_mid_wgs_fp8 : bf16_max);
}

// Split-K for the 256-row fp8 tile kernel in dense mode: with ceil(M / 256) x N / 256 workgroups well under the CU count the
// launch takes one tile's time on a part of the chip; K ranges (fp32 partials, ordered reduce) fill it.  Ranges are whole
// 128-wide blocks, at least sixteen per range, as many ranges as keep the workgroup count within ~1.25 x the CUs.
int tuned_fp8_ksplit(int M, int N, int K, int fill = 2) {   // split only when the tiles fill at most 1 / fill of the chip
    if (knobs().no_tuned_splitk || N % 256 != 0 || K % 128 != 0) return 1;
    const int64_t wgs = ceil_div(M, 256) * (int64_t)(N / 256);
    const int cus = device_cu_count(), kb = K >> 7;
    if (wgs * fill > cus) return 1;
    int best = 1;
    for (int ks = 2; ks <= 8; ++ks) {
        if (kb % ks != 0 || kb / ks < 16) continue;   // shorter ranges lose to the fp32 partial round trip (4096 x 1536 x 2048: 0.051 -> 0.056 ms)
        if (wgs * ks > cus + cus / 4) break;
        if ((int64_t)ks * M * N * 4 > (256ll << 20)) break;
        best = ks;
    }
    return best;
}

void fill_tuned(MoeGemmParams& g, const void* x, int64_t x_stride, int M, const int* ident, const void* w,
                const float* scale, int R, int C, int block_n, const int4* tile_info, const int* num_tiles) {
    g.x = (const uint16_t*)x;
    g.x_stride = x_stride;
    g.x_bytes = (int64_t)M * x_stride * 2;
    g.sorted_slot = ident;
    g.topk = 1;
    g.w = (const uint8_t*)w;
    g.w_expert_stride = (int64_t)R * C;
    g.w_scale = scale;
    g.scale_rows = (int)ceil_div(R, block_n);
    g.scale_cols = C / 128;
    g.block_n = block_n;
    g.C = C;
    g.tile_info = tile_info;
    g.num_tiles = num_tiles;
}

DenseWs plan_dense(int M, int N, int K, bool need_ic1, bool int8_act, bool fp8_split = false) {
    DenseWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes ? bytes : 1, 256);
        return o;
    };
    const int tiles = (int)ceil_div(M > 0 ? M : 1, kGenericTileM);
    w.tile_info = take((size_t)tiles * 16);
    w.num_tiles = take(4);
    w.ident = take((size_t)(M > 0 ? M : 1) * 4);
    if (need_ic1) w.ic1 = take((size_t)M * N * (int8_act ? 4 : 2));   // W8A8 keeps SiLU*mul in fp32 until quantised
    if (int8_act) {
        w.xq = take((size_t)M * K);
        w.xs = take((size_t)M * 4);
        if (need_ic1) {
            w.ic1q = take((size_t)M * N);
            w.ic1s = take((size_t)M * 4);
        }
    }
    if (need_ic1) {    // shared expert at decode sizes on the weight-streaming kernel: partials of both GEMMs share one buffer
        int k1 = mid_dense_ksplit(M, 2 * N, K), k2 = mid_dense_ksplit(M, K, N);
        const int j1 = bf16_mid_ksplit(M, 2 * N, K), j2 = bf16_mid_ksplit(M, K, N);    // bf16 weights: same scheme
        if (j1 > k1) k1 = j1;
        if (j2 > k2) k2 = j2;
        const int i1 = shared_i8_ksplit(M, 2 * N, K, N), i2 = shared_i8_ksplit(M, K, N, N);      // int8 weights: int32 partials, same size
        if (i1 > k1) k1 = i1;
        if (i2 > k2) k2 = i2;
        if (k1 >= 1 && k2 >= 1) {
            const size_t b1 = (size_t)k1 * M * 2 * N * 4, b2 = (size_t)k2 * M * K * 4;
            w.partial = take(b1 > b2 ? b1 : b2);
        }
    }
    if (fp8_split && !need_ic1 && dense_s128_shape_ok(M, N, K)) {   // fp8 W8A16 on the two-term kernel: x as (hi, lo) e4m3 + scale bytes
        w.xsplit = take((size_t)M * 2 * K);
        w.xsplit_s = take((size_t)M * align_up(K / 128, 4));
    }
    if (!need_ic1) {   // plain dense GEMM: fp32 partials of the split-K form (small M)
        int ks = generic_ksplit(M, N, K);
        const int km = mid_dense_ksplit(M, N, K);          // fp8 / bf16 small-M paths (largest of the plans)
        if (km > ks) ks = km;
        const int kb = bf16_mid_ksplit(M, N, K);
        if (kb > ks) ks = kb;
        const int ki = i8_mid_dense_ksplit(M, N, K);
        if (ki > ks) ks = ki;
        if (M >= 192) {                                     // 256-row fp8 tile kernel with K ranges
            const int kt = tuned_fp8_ksplit(M, N, K);
            if (kt > ks) ks = kt;
        }
        if (ks > 1) w.partial = take((size_t)ks * M * N * 4);
    }
    w.total = off;
    return w;
}

void fill_weight(GenericGemmParams& g, const void* w, const float* scale, int wtype, int packed, int rows, int C,
                 int block_n) {
    g.w = w;
    g.w_type = wtype;
    g.packed = packed;
    g.w_expert_stride = 0;
    g.C = C;
    g.w_scale = scale;
    if (wtype == SGLK_W_FP8_E4M3) {
        g.scale_rows = (int)ceil_div(rows, block_n);
        g.scale_cols = (int)ceil_div(C, 128);
        g.block_n = block_n;
    } else {
        g.scale_rows = rows;
        g.scale_cols = 1;
        g.block_n = 1;
    }
}

int check_weight(const char* op, int wtype, int packed, int rows, int C, const float* scale, int block_n, int block_k) {
    SGLK_REQUIRE(wtype == SGLK_W_BF16 || wtype == SGLK_W_FP8_E4M3 || wtype == SGLK_W_INT8, SGLK_ERR_INVALID,
                 "%s: unknown weight type %d", op, wtype);
    if (wtype == SGLK_W_FP8_E4M3) {
        SGLK_REQUIRE(scale, SGLK_ERR_INVALID, "%s: fp8 weights need block scales", op);
        SGLK_REQUIRE(block_k == 128, SGLK_ERR_SHAPE, "%s: block_size[1] must be 128 (got %d)", op, block_k);
        SGLK_REQUIRE(block_n > 0 && block_n % 16 == 0, SGLK_ERR_SHAPE, "%s: block_size[0] must be a multiple of 16 (got %d)", op, block_n);
    } else if (wtype == SGLK_W_INT8) {
        SGLK_REQUIRE(scale, SGLK_ERR_INVALID, "%s: int8 weights need per-channel scales", op);
    }
    if (packed) {
        const bool ok = wtype == SGLK_W_BF16 ? (rows % 32 == 0 && C % 8 == 0) : (rows % 16 == 0 && C % 64 == 0);
        SGLK_REQUIRE(ok, SGLK_ERR_SHAPE, "%s: a [%d][%d] weight cannot be in packed order; pass packed=0", op, rows, C);
    }
    return SGLK_OK;
}

}  // namespace

extern "C" size_t sglk_shared_expert_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t wtype) {
    if (M < 0 || N <= 0 || K <= 0) return 0;
    return plan_dense(M, N, K, true, wtype == SGLK_W_INT8).total;
}

namespace {
// Row-major shared-expert weights (the reference's own 12-argument call, /root/reference/test_shared_experts.py:68), at EVERY
// size: re-tile both into the workspace (one pass over their bytes) and run the packed paths -- 3-5x over the generic engine at
// prefill sizes, and still 3x at M = 1 (the generic engine streamed the 88 MB of a 2048 x 7168 bf16 expert at 0.5 TB/s).
bool shared_pack_on_the_fly(int M, int N, int K, int wtype, int packed) {
    if (packed || M < 1 || knobs().force_generic || knobs().no_pack_on_the_fly) return false;   // M == 0: nothing to run
    if (wtype == SGLK_W_BF16) return (2 * N) % 32 == 0 && K % 32 == 0 && N % 8 == 0;
    return (wtype == SGLK_W_FP8_E4M3 || wtype == SGLK_W_INT8) && (2 * N) % 16 == 0 && K % 64 == 0 && N % 64 == 0;
}
}  // namespace

extern "C" size_t sglk_shared_expert_workspace_bytes_ex(int32_t M, int32_t N, int32_t K, int32_t wtype, int32_t packed) {
    if (M < 0 || N <= 0 || K <= 0) return 0;
    const size_t base = align_up(plan_dense(M, N, K, true, wtype == SGLK_W_INT8).total, 256);
    return shared_pack_on_the_fly(M, N, K, wtype, packed) ? base + align_up((size_t)3 * N * K * (wtype == SGLK_W_BF16 ? 2 : 1), 256) : base;
}

namespace sglk {
// the decode-size fp8 path of shared_expert_impl (two split-K passes of the weight-streaming kernel), the one whose last
// launch can take the routed experts' combine as a slot addend
bool shared_expert_can_fold(const sglk_shared_expert_args* a) {
    return a->wtype == SGLK_W_FP8_E4M3 && (a->packed & 3) == 3 && a->block_k == 128 && a->block_n > 0 && a->block_n % 16 == 0 &&
           a->hidden_stride % 8 == 0 && ((uintptr_t)a->hidden % 16) == 0 && !knobs().force_generic && a->N % 128 == 0 &&
           !tuned_dense_ok(a->M, 2 * a->N, a->K, a->wtype, 1, a->block_n, a->block_k, a->hidden, a->hidden_stride) &&
           mid_dense_ksplit(a->M, 2 * a->N, a->K) >= 1 && mid_dense_ksplit(a->M, a->K, a->N) >= 1;
}

int shared_expert_impl(const sglk_shared_expert_args* a, void* stream, const MoeSlotAddend* moe);
}  // namespace sglk

extern "C" int sglk_shared_expert(const sglk_shared_expert_args* a, void* stream) { return sglk::shared_expert_impl(a, stream, nullptr); }

int sglk::shared_expert_impl(const sglk_shared_expert_args* a, void* stream, const MoeSlotAddend* moe) {
    SGLK_REQUIRE(a, SGLK_ERR_INVALID, "shared_expert: null args");
    const int M = a->M, N = a->N, K = a->K;
    SGLK_REQUIRE(M >= 0 && N > 0 && K > 0, SGLK_ERR_INVALID, "shared_expert: bad sizes M=%d N=%d K=%d", M, N, K);
    SGLK_REQUIRE(a->w1 && a->w2 && a->workspace, SGLK_ERR_INVALID, "shared_expert: null pointer");
    SGLK_REQUIRE(M == 0 || (a->hidden && a->out && (a->fused_out || moe)), SGLK_ERR_INVALID, "shared_expert: null pointer");
    SGLK_REQUIRE(!moe || shared_expert_can_fold(a), SGLK_ERR_INVALID, "shared_expert: this path cannot fold the routed combine");
    SGLK_REQUIRE(a->hidden_stride >= K && a->out_stride >= K && (moe || a->fused_out_stride >= K), SGLK_ERR_INVALID,
                 "shared_expert: row stride < K");
    int rc = check_weight("shared_expert", a->wtype, a->packed & 1, 2 * N, K, a->w1_scale, a->block_n, a->block_k);
    if (rc != SGLK_OK) return rc;
    rc = check_weight("shared_expert", a->wtype, (a->packed >> 1) & 1, K, N, a->w2_scale, a->block_n, a->block_k);
    if (rc != SGLK_OK) return rc;
    if (a->wtype == SGLK_W_FP8_E4M3) SGLK_REQUIRE(N % 16 == 0, SGLK_ERR_SHAPE, "shared_expert(fp8): N (%d) must be a multiple of 16", N);
    const bool i8 = a->wtype == SGLK_W_INT8;
    const DenseWs w = plan_dense(M, N, K, true, i8);
    SGLK_REQUIRE(a->workspace_bytes >= w.total, SGLK_ERR_WORKSPACE, "shared_expert: workspace %zu < required %zu",
                 a->workspace_bytes, w.total);
    if (M == 0) return SGLK_OK;
    hipStream_t s = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)a->workspace;
    // row-major weights, prefill-size M, and the caller sized the workspace with _ex: re-tile into the workspace, run as packed
    if (!moe && shared_pack_on_the_fly(M, N, K, a->wtype, a->packed)) {
        const size_t base = align_up(w.total, 256), elt = a->wtype == SGLK_W_BF16 ? 2 : 1;
        if (a->workspace_bytes >= base + (size_t)3 * N * K * elt) {
            unsigned char* p1 = ws + base;
            unsigned char* p2 = p1 + (size_t)2 * N * K * elt;
            rc = sglk_pack_weight(a->w1, p1, 1, 2 * N, K, a->wtype, stream);
            if (rc != SGLK_OK) return rc;
            rc = sglk_pack_weight(a->w2, p2, 1, K, N, a->wtype, stream);
            if (rc != SGLK_OK) return rc;
            sglk_shared_expert_args b = *a;
            b.w1 = p1;
            b.w2 = p2;
            b.packed = 3;
            b.workspace_bytes = base;
            return shared_expert_impl(&b, stream, nullptr);
        }
    }
    int4* tile_info = (int4*)(ws + w.tile_info);
    int* num_tiles = (int*)(ws + w.num_tiles);
    uint16_t* ic1 = (uint16_t*)(ws + w.ic1);
    // (up to SGLK_SHARED_MID_MAX rows the split-K passes below are taken even where the tile kernel could run)
    const bool shared_mid = shared_rows_take_mid(M, N, knobs().shared_mid_max) && a->wtype == SGLK_W_FP8_E4M3 && a->block_k == 128 && a->block_n > 0 &&
                            a->block_n % 16 == 0 && a->hidden_stride % 8 == 0 && ((uintptr_t)a->hidden % 16) == 0 &&
                            mid_dense_ksplit(M, 2 * N, K) >= 1 && mid_dense_ksplit(M, K, N) >= 1;
    if ((a->packed & 3) == 3 && N % 128 == 0 && !shared_mid &&
        tuned_dense_ok(M, K, N, a->wtype, 1, a->block_n, a->block_k, ic1, N) &&
        tuned_dense_ok(M, 2 * N, K, a->wtype, 1, a->block_n, a->block_k, a->hidden, a->hidden_stride) &&
        a->out_stride % 8 == 0 && ((uintptr_t)a->out % 16) == 0 && a->fused_out_stride % 4 == 0) {
        // large fp8 shared expert on the tuned grouped-GEMM kernel (one "expert", identity row map)
        int* ident = (int*)(ws + w.ident);
        const int t256 = (int)ceil_div(M, 256);
        rc = launch_dense_tiles(M, 256, tile_info, num_tiles, ident, s);
        if (rc != SGLK_OK) return rc;
        MoeGemmParams t1{};
        fill_tuned(t1, a->hidden, a->hidden_stride, M, ident, a->w1, a->w1_scale, 2 * N, K, a->block_n, tile_info, num_tiles);
        t1.n_half = N;
        t1.n_tiles = N / 128;
        t1.out = ic1;
        t1.out_stride = N;
        rc = launch_moe_gemm_fp8w_256i(MODE_GATE_UP, t1, t256, s);
        if (rc != SGLK_OK) return rc;
        MoeGemmParams t2{};
        fill_tuned(t2, ic1, N, M, ident, a->w2, a->w2_scale, K, N, a->block_n, tile_info, num_tiles);
        t2.n_tiles = K / 256;
        t2.out = (uint16_t*)a->out;
        t2.out_stride = a->out_stride;
        t2.addend = (const uint16_t*)a->fused_out;
        t2.addend_stride = a->fused_out_stride;
        t2.addend_scale = a->routed_scaling_factor;
        return launch_moe_gemm_fp8w_256i(MODE_PLAIN, t2, t256, s);
    }
    // fp8, decode-size M: both GEMMs as split-K passes of the weight-streaming kernel; the gate_up partials are reduced by a
    // SiLU*mul pass (fp32 until the single bf16 rounding of ic1, like the fused path), the down partials by the ordered reduce
    // that also adds fused_out * routed_scaling_factor
    if (a->wtype == SGLK_W_FP8_E4M3 && (a->packed & 3) == 3 && a->block_k == 128 && a->block_n > 0 && a->block_n % 16 == 0 &&
        a->hidden_stride % 8 == 0 && ((uintptr_t)a->hidden % 16) == 0 && !knobs().force_generic) {
        const int k1 = mid_dense_ksplit(M, 2 * N, K), k2 = mid_dense_ksplit(M, K, N);
        if (k1 >= 1 && k2 >= 1) {
            float* partial = (float*)(ws + w.partial);
            const int mt = (int)ceil_div(M, kMidTileM);
            MoeGemmParams t1{};
            fill_tuned(t1, a->hidden, a->hidden_stride, M, nullptr, a->w1, a->w1_scale, 2 * N, K, a->block_n, nullptr, nullptr);
            t1.n_tiles = 2 * N / 128;
            t1.ksplit = k1;
            t1.split_kblocks = (K >> 7) / k1;
            t1.split_rows = M;
            t1.out_cols = 2 * N;
            t1.partial = partial;
            rc = launch_moe_gemm_fp8w_mid(MODE_PLAIN, t1, mt, s);
            if (rc != SGLK_OK) return rc;
            rc = launch_splitk_reduce_silu_mul(partial, k1, M, N, ic1, N, s);
            if (rc != SGLK_OK) return rc;
            MoeGemmParams t2{};
            fill_tuned(t2, ic1, N, M, nullptr, a->w2, a->w2_scale, K, N, a->block_n, nullptr, nullptr);
            t2.n_tiles = K / 128;
            t2.ksplit = k2;
            t2.split_kblocks = (N >> 7) / k2;
            t2.split_rows = M;
            t2.out_cols = K;
            t2.partial = partial;
            rc = launch_moe_gemm_fp8w_mid(MODE_PLAIN, t2, mt, s);
            if (rc != SGLK_OK) return rc;
            GenericGemmParams r{};
            r.partial = partial;
            r.ksplit = k2;
            r.split_rows = M;
            r.n_out = K;
            r.out = a->out;
            r.out_type = SGLK_OUT_BF16;
            r.out_stride = a->out_stride;
            r.addend_scale = a->routed_scaling_factor;
            if (moe) {
                r.moe_ic2 = moe->ic2;
                r.moe_ids = moe->topk_ids;
                r.moe_topk = moe->topk;
                r.moe_E = moe->E;
            } else {
                r.addend = a->fused_out;
                r.addend_stride = a->fused_out_stride;
            }
            return launch_splitk_reduce(r, s);
        }
    }
    // bf16 / int8 packed weights at prefill sizes: the tuned 256-row kernels of fused_experts -- gate_up + SiLU*mul in the grouped form with
    // one "expert" and an identity row map, down in the dense form with fused_out * routed_scaling_factor added in fp32 before the
    // single bf16 rounding.  (The generic engine took 0.71 ms (bf16) / 0.66 ms (int8) at 2048 x 2048 x 7168.)  Below
    // SGLK_SHARED_MID_MAX rows the split-K passes further down are taken where they exist (their crossover is the fp8 one's).
    const bool big_common = (a->packed & 3) == 3 && !moe && !knobs().force_generic && M >= 192 && N % 128 == 0 && K % 256 == 0 &&
                            a->out_stride % 8 == 0 && ((uintptr_t)a->out % 16) == 0 && a->fused_out_stride % 4 == 0 &&
                            ((uintptr_t)a->fused_out % 8) == 0 && (int64_t)4 * N * K < (1ll << 32) && (int64_t)M * N * 4 < (1ll << 32);
    if (big_common && a->wtype == SGLK_W_BF16 && a->hidden_stride % 8 == 0 && ((uintptr_t)a->hidden % 16) == 0 &&
        (int64_t)M * a->hidden_stride * 2 < (1ll << 32) &&
        (!shared_rows_take_mid(M, N, knobs().shared_mid_max) || bf16_mid_ksplit(M, 2 * N, K) < 1 || bf16_mid_ksplit(M, K, N) < 1)) {
        int* ident = (int*)(ws + w.ident);
        const int t256 = (int)ceil_div(M, 256);
        rc = launch_dense_tiles(M, 256, tile_info, num_tiles, ident, s);
        if (rc != SGLK_OK) return rc;
        Bf16GemmParams q1{};
        q1.x = (const uint16_t*)a->hidden;
        q1.x_stride = a->hidden_stride * 2;
        q1.x_bytes = (int64_t)M * a->hidden_stride * 2;
        q1.w = (const uint8_t*)a->w1;
        q1.w_bytes = (int64_t)2 * N * K * 2;
        q1.out = ic1;
        q1.out_stride = N;
        q1.M = M;
        q1.K = K;
        q1.n_tiles = N / 128;
        q1.tile_info = tile_info;
        q1.num_tiles = num_tiles;
        q1.sorted_slot = ident;
        q1.topk = 1;
        q1.n_half = N;
        rc = launch_gemm_bf16_256(MODE_GATE_UP, q1, t256, s);
        if (rc != SGLK_OK) return rc;
        Bf16GemmParams q2{};
        q2.x = ic1;
        q2.x_stride = (int64_t)N * 2;
        q2.x_bytes = (int64_t)M * N * 2;
        q2.w = (const uint8_t*)a->w2;
        q2.w_bytes = (int64_t)K * N * 2;
        q2.out = (uint16_t*)a->out;
        q2.out_stride = a->out_stride;
        q2.M = M;
        q2.K = N;
        q2.n_tiles = K / 256;
        q2.addend = (const uint16_t*)a->fused_out;
        q2.addend_stride = a->fused_out_stride;
        q2.addend_scale = a->routed_scaling_factor;
        return launch_gemm_bf16_256(MODE_PLAIN, q2, t256, s);
    }
    if (big_common && i8 && N >= 256 && (int64_t)M * K < (1ll << 32) &&
        (shared_i8_ksplit(M, 2 * N, K, N) < 1 || shared_i8_ksplit(M, K, N, N) < 1 || knobs().no_i8_mid)) {
        int* ident = (int*)(ws + w.ident);
        const int t256 = (int)ceil_div(M, 256);
        rc = launch_dense_tiles(M, 256, tile_info, num_tiles, ident, s);
        if (rc != SGLK_OK) return rc;
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
        q1.out = ic1;                  // fp32 [row][N]: SiLU*mul stays in fp32 until it is quantised
        q1.out_stride = N;
        q1.M = M;
        q1.K = K;
        q1.n_tiles = N / 128;
        q1.tile_info = tile_info;
        q1.num_tiles = num_tiles;
        q1.sorted_slot = ident;
        q1.topk = 1;
        q1.n_half = N;
        rc = launch_gemm_i8_256(MODE_GATE_UP, q1, t256, s);
        if (rc != SGLK_OK) return rc;
        int8_t* hq = (int8_t*)(ws + w.ic1q);
        float* hs = (float*)(ws + w.ic1s);
        rc = launch_quant_int8_rows_f32((const float*)ic1, N, hq, N, hs, M, N, 1e-7f, s);
        if (rc != SGLK_OK) return rc;
        I8GemmParams q2{};
        q2.x = hq;
        q2.x_stride = N;
        q2.x_bytes = (int64_t)M * N;
        q2.x_scale = hs;
        q2.w = (const uint8_t*)a->w2;
        q2.w_bytes = (int64_t)K * N;
        q2.w_scale = a->w2_scale;
        q2.scale_rows = K;
        q2.out = (uint16_t*)a->out;
        q2.out_stride = a->out_stride;
        q2.M = M;
        q2.K = N;
        q2.n_tiles = K / 256;
        q2.addend = (const uint16_t*)a->fused_out;
        q2.addend_stride = a->fused_out_stride;
        q2.addend_scale = a->routed_scaling_factor;
        return launch_gemm_i8_256(MODE_PLAIN, q2, t256, s);
    }
    // int8 W8A8, decode sizes and up to SGLK_SHARED_I8_MID_MAX rows (shared_i8_ksplit): quantise x, gate_up as exact int32 split-K partials, reduce with the scales + SiLU*mul
    // (fp32 ic1), quantise ic1, down as int32 partials, reduce with the scales + fused_out * routed_scaling_factor
    if (i8 && (a->packed & 3) == 3 && !knobs().force_generic && !knobs().no_i8_mid) {
        const int k1 = shared_i8_ksplit(M, 2 * N, K, N), k2 = shared_i8_ksplit(M, K, N, N);
        if (k1 >= 1 && k2 >= 1) {
            int32_t* partial = (int32_t*)(ws + w.partial);
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
            q1.M = M; q1.N = 2 * N; q1.K = K;
            q1.ksplit = k1;
            q1.split_kblocks = (K >> 7) / k1;
            q1.partial_i32 = partial;
            q1.out = nullptr;
            rc = launch_gemm_i8_mid_plain(q1, s);
            if (rc != SGLK_OK) return rc;
            rc = launch_i8_reduce_silu_mul(partial, k1, M, N, xs, a->w1_scale, (float*)ic1, s);
            if (rc != SGLK_OK) return rc;
            int8_t* hq = (int8_t*)(ws + w.ic1q);
            float* hs = (float*)(ws + w.ic1s);
            rc = launch_quant_int8_rows_f32((const float*)ic1, N, hq, N, hs, M, N, 1e-7f, s);
            if (rc != SGLK_OK) return rc;
            I8GemmParams q2{};
            q2.x = hq;
            q2.x_stride = N;
            q2.x_scale = hs;
            q2.w = (const uint8_t*)a->w2;
            q2.w_bytes = (int64_t)K * N;
            q2.w_scale = a->w2_scale;
            q2.M = M; q2.N = K; q2.K = N;
            q2.ksplit = k2;
            q2.split_kblocks = (N >> 7) / k2;
            q2.partial_i32 = partial;
            q2.out = nullptr;
            rc = launch_gemm_i8_mid_plain(q2, s);
            if (rc != SGLK_OK) return rc;
            return launch_i8_reduce_addend(partial, k2, M, K, hs, a->w2_scale, (const uint16_t*)a->fused_out, a->fused_out_stride,
                                           a->routed_scaling_factor, (uint16_t*)a->out, a->out_stride, s);
        }
    }
    // bf16 packed weights, decode sizes: the same four launches on the bf16 weight-streaming kernel
    if (a->wtype == SGLK_W_BF16 && (a->packed & 3) == 3 && a->hidden_stride % 8 == 0 && ((uintptr_t)a->hidden % 16) == 0 &&
        !knobs().force_generic) {
        const int k1 = bf16_mid_ksplit(M, 2 * N, K), k2 = bf16_mid_ksplit(M, K, N);
        if (k1 >= 1 && k2 >= 1) {
            float* partial = (float*)(ws + w.partial);
            BmidParams q1{};
            q1.x = (const uint16_t*)a->hidden;
            q1.x_stride = a->hidden_stride;
            q1.w = (const uint8_t*)a->w1;
            q1.M = M; q1.N = 2 * N; q1.K = K;
            q1.ksplit = k1;
            q1.split_kblocks = (K >> 7) / k1;
            q1.partial = partial;
            rc = launch_gemm_bf16_mid(q1, s);
            if (rc != SGLK_OK) return rc;
            rc = launch_splitk_reduce_silu_mul(partial, k1, M, N, ic1, N, s);
            if (rc != SGLK_OK) return rc;
            BmidParams q2{};
            q2.x = ic1;
            q2.x_stride = N;
            q2.w = (const uint8_t*)a->w2;
            q2.M = M; q2.N = K; q2.K = N;
            q2.ksplit = k2;
            q2.split_kblocks = (N >> 7) / k2;
            q2.partial = partial;
            rc = launch_gemm_bf16_mid(q2, s);
            if (rc != SGLK_OK) return rc;
            GenericGemmParams r{};
            r.partial = partial;
            r.ksplit = k2;
            r.split_rows = M;
            r.n_out = K;
            r.out = a->out;
            r.out_type = SGLK_OUT_BF16;
            r.out_stride = a->out_stride;
            r.addend = a->fused_out;
            r.addend_stride = a->fused_out_stride;
            r.addend_scale = a->routed_scaling_factor;
            return launch_splitk_reduce(r, s);
        }
    }
    const int tiles = (int)ceil_div(M, kGenericTileM);
    rc = launch_dense_tiles(M, kGenericTileM, tile_info, num_tiles, nullptr, s);
    if (rc != SGLK_OK) return rc;

    GenericGemmParams g1{};
    g1.x = a->hidden;
    g1.x_type = SGLK_W_BF16;
    g1.x_stride = a->hidden_stride;
    if (i8) {
        rc = launch_quant_int8_rows((const uint16_t*)a->hidden, a->hidden_stride, (int8_t*)(ws + w.xq), K,
                                    (float*)(ws + w.xs), M, K, 1e-7f, s);
        if (rc != SGLK_OK) return rc;
        g1.x = ws + w.xq;
        g1.x_type = SGLK_W_INT8;
        g1.x_stride = K;
        g1.x_row_scale = (const float*)(ws + w.xs);
    }
    g1.topk = 1;
    g1.gather = GG_GATHER_NONE;
    g1.tile_info = tile_info;
    g1.num_tiles = num_tiles;
    g1.n_tiles = (int)ceil_div(N, 32);
    fill_weight(g1, a->w1, a->w1_scale, a->wtype, a->packed & 1, 2 * N, K, a->block_n);
    g1.n_half = N;
    g1.n_out = N;
    g1.out = ic1;
    g1.out_type = i8 ? SGLK_OUT_F32 : SGLK_OUT_BF16;
    g1.out_stride = N;
    rc = launch_gemm_generic(GG_GATE_UP, g1, tiles, s);
    if (rc != SGLK_OK) return rc;

    GenericGemmParams g2{};
    g2.x = ic1;
    g2.x_type = SGLK_W_BF16;
    g2.x_stride = N;
    if (i8) {
        rc = launch_quant_int8_rows_f32((const float*)ic1, N, (int8_t*)(ws + w.ic1q), N, (float*)(ws + w.ic1s), M, N,
                                        1e-7f, s);
        if (rc != SGLK_OK) return rc;
        g2.x = ws + w.ic1q;
        g2.x_type = SGLK_W_INT8;
        g2.x_row_scale = (const float*)(ws + w.ic1s);
    }
    g2.topk = 1;
    g2.gather = GG_GATHER_NONE;
    g2.tile_info = tile_info;
    g2.num_tiles = num_tiles;
    g2.n_tiles = (int)ceil_div(K, 64);
    fill_weight(g2, a->w2, a->w2_scale, a->wtype, (a->packed >> 1) & 1, K, N, a->block_n);
    g2.n_out = K;
    g2.out = a->out;
    g2.out_type = SGLK_OUT_BF16;
    g2.out_stride = a->out_stride;
    g2.addend = a->fused_out;
    g2.addend_stride = a->fused_out_stride;
    g2.addend_scale = a->routed_scaling_factor;
    return launch_gemm_generic(GG_PLAIN, g2, tiles, s);
}

extern "C" size_t sglk_scaled_mm_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t wtype, int32_t x_is_int8) {
    if (M < 0 || N <= 0 || K <= 0) return 0;
    return plan_dense(M, N, K, false, wtype == SGLK_W_INT8 && !x_is_int8, wtype == SGLK_W_FP8_E4M3 && !x_is_int8).total;
}

namespace sglk {
// Row-major weights at prefill sizes: re-tiling the weight into the workspace (one pass over its bytes) and running the tuned
// kernel beats the generic engine 3-4x (1000 x 18432 x 2560 int8: 0.32 -> 0.09 ms); decode sizes are bound by the weight bytes
// and read them once either way.  Shapes sglk_pack_weight takes: bf16 rows % 32 / cols % 8, fp8 / int8 rows % 16 / cols % 64.
static bool pack_on_the_fly(int M, int N, int K, int wtype, int packed) {
    if (packed || knobs().force_generic || knobs().no_pack_on_the_fly) return false;
    int min_rows = knobs().pack_min_rows;
    if (min_rows <= 0) {
        // Below 192 rows the re-tiling pass (one read + one write of the weight) still pays where the generic engine is far from
        // reading the weight once: same-box A/B (tools/ab_pack_min_rows.py, profiles/r03_ab_pack_min_rows.txt), 191 x 4096 x 4096:
        // fp8 53 -> 28 us, int8 with quantisation 68 -> 32, bf16 51 -> 37; crossover at 64 rows (fp8, int8) / 128 rows (bf16) for
        // layers of >= 4 Mi elements, smaller bf16 / fp8 layers stay on the generic engine (512 x 1024: 11 vs 14 us)
        const bool big = (int64_t)N * K >= (1ll << 22);
        min_rows = wtype == SGLK_W_INT8 ? 64 : (!big ? 192 : (wtype == SGLK_W_BF16 ? 128 : 64));
    }
    if (M < min_rows) return false;
    if (wtype == SGLK_W_BF16) return N % 32 == 0 && K % 8 == 0;
    return (wtype == SGLK_W_FP8_E4M3 || wtype == SGLK_W_INT8) && N % 16 == 0 && K % 64 == 0;
}
static size_t weight_bytes(int N, int K, int wtype) { return (size_t)N * K * (wtype == SGLK_W_BF16 ? 2 : 1); }
}  // namespace sglk

extern "C" size_t sglk_scaled_mm_workspace_bytes_ex(int32_t M, int32_t N, int32_t K, int32_t wtype, int32_t x_is_int8,
                                                    int32_t packed) {
    if (M < 0 || N <= 0 || K <= 0) return 0;
    const size_t base = align_up(plan_dense(M, N, K, false, wtype == SGLK_W_INT8 && !x_is_int8, wtype == SGLK_W_FP8_E4M3 && !x_is_int8).total, 256);
    return pack_on_the_fly(M, N, K, wtype, packed) ? base + align_up(weight_bytes(N, K, wtype), 256) : base;
}

extern "C" int sglk_scaled_mm(const sglk_scaled_mm_args* a, void* stream) {
    SGLK_REQUIRE(a, SGLK_ERR_INVALID, "scaled_mm: null args");
    const int M = a->M, N = a->N, K = a->K;
    SGLK_REQUIRE(M >= 0 && N > 0 && K > 0, SGLK_ERR_INVALID, "scaled_mm: bad sizes M=%d N=%d K=%d", M, N, K);
    SGLK_REQUIRE(a->w && a->workspace, SGLK_ERR_INVALID, "scaled_mm: null pointer");
    SGLK_REQUIRE(M == 0 || (a->x && a->out), SGLK_ERR_INVALID, "scaled_mm: null pointer");
    SGLK_REQUIRE(a->x_stride >= K && a->out_stride >= N, SGLK_ERR_INVALID, "scaled_mm: row stride too small");
    SGLK_REQUIRE(a->out_type >= SGLK_OUT_BF16 && a->out_type <= SGLK_OUT_F32, SGLK_ERR_INVALID, "scaled_mm: bad out_type");
    int rc = check_weight("scaled_mm", a->wtype, a->packed, N, K, a->w_scale, a->block_n, a->block_k);
    if (rc != SGLK_OK) return rc;
    if (a->x_is_int8) {
        SGLK_REQUIRE(a->wtype == SGLK_W_INT8 && a->x_scale, SGLK_ERR_INVALID,
                     "scaled_mm: int8 activations need int8 weights and x_scale");
    }
    const bool quant_here = a->wtype == SGLK_W_INT8 && !a->x_is_int8;
    const DenseWs w = plan_dense(M, N, K, false, quant_here, a->wtype == SGLK_W_FP8_E4M3 && !a->x_is_int8);
    SGLK_REQUIRE(a->workspace_bytes >= w.total, SGLK_ERR_WORKSPACE, "scaled_mm: workspace %zu < required %zu",
                 a->workspace_bytes, w.total);
    if (M == 0) return SGLK_OK;
    hipStream_t s = (hipStream_t)stream;
    unsigned char* ws = (unsigned char*)a->workspace;
    // row-major weight, prefill-size M, and the caller sized the workspace with _ex: re-tile into the workspace, run as packed
    if (pack_on_the_fly(M, N, K, a->wtype, a->packed)) {
        const size_t base = align_up(w.total, 256);
        if (a->workspace_bytes >= base + weight_bytes(N, K, a->wtype)) {
            rc = sglk_pack_weight(a->w, ws + base, 1, N, K, a->wtype, stream);
            if (rc != SGLK_OK) return rc;
            sglk_scaled_mm_args b = *a;
            b.w = ws + base;
            b.packed = 1;
            b.workspace_bytes = base;
            return sglk_scaled_mm(&b, stream);
        }
    }
    int4* tile_info = (int4*)(ws + w.tile_info);
    int* num_tiles = (int*)(ws + w.num_tiles);
    if (a->out_type == SGLK_OUT_BF16 && !a->x_is_int8 && a->out_stride % 8 == 0 && ((uintptr_t)a->out % 16) == 0 &&
        (!a->bias || ((uintptr_t)a->bias % 16) == 0) &&
        tuned_dense_ok(M, N, K, a->wtype, a->packed, a->block_n, a->block_k, a->x, a->x_stride) &&
        !(dense_prefers_mid(M, N, a->wtype) && mid_dense_ksplit(M, N, K) >= 1 && a->block_n % 16 == 0)) {
        int* ident = (int*)(ws + w.ident);
        const int t256 = (int)ceil_div(M, 256);
        const int kt = tuned_fp8_ksplit(M, N, K);
        // enough rows to fill the chip: the two-term e4m3 split on the scaled fp8 matrix cores, 128-row tiles, two workgroups per CU
        // (moe_gemm_fp8w_s128.hip, MODE_PLAIN) -- the W8A16 contract with exact products, as in fused_experts
        if (kt == 1 && w.xsplit && dense_s128_shape_ok(M, N, K) && (int64_t)M * 2 * K < (1ll << 32) && a->block_n % 32 == 0) {
            const int t128 = (int)ceil_div(M, 128);
            uint8_t* xq = ws + w.xsplit;
            uint8_t* xs = ws + w.xsplit_s;
            const int xs_stride = (int)align_up(K / 128, 4);
            rc = launch_split_fp8_block128((const uint16_t*)a->x, a->x_stride, xq, 2 * (int64_t)K, xs, xs_stride, M, K, s);
            if (rc != SGLK_OK) return rc;
            A8GemmParams q{};
            q.x = xq;
            q.x_stride = 2 * (int64_t)K;
            q.x_bytes = (int64_t)M * 2 * K;
            q.xs = xs;
            q.xs_stride = xs_stride;
            q.topk = 1;
            q.w = (const uint8_t*)a->w;
            q.w_expert_stride = (int64_t)N * K;
            q.w_scale = a->w_scale;
            q.scale_rows = (int)ceil_div(N, a->block_n);
            q.scale_cols = K / 128;
            q.block_n = a->block_n;
            q.C = K;
            q.dense_rows = M;
            q.n_tiles = N / 256;
            q.out = a->out;
            q.out_stride = a->out_stride;
            q.bias = a->bias;
            q.max_mtiles = t128;
            return launch_moe_gemm_fp8w_s128(MODE_PLAIN, q, t128, s, 2);
        }
        if (kt > 1 && w.partial) {   // K ranges as the tile table's "experts", fp32 partials, ordered reduce (+ bias)
            rc = launch_dense_tiles_ksplit(M, 256, kt, tile_info, num_tiles, ident, s);
            if (rc != SGLK_OK) return rc;
            const int kr = K / kt;
            MoeGemmParams t{};
            fill_tuned(t, a->x, a->x_stride, M, ident, a->w, a->w_scale, N, K, a->block_n, tile_info, num_tiles);
            t.C = kr;
            t.c_full = K;
            t.w_expert_stride = (int64_t)(kr >> 6) * 1024;
            t.w_bytes_total = (int64_t)N * K;
            t.n_tiles = N / 256;
            t.ksplit = kt;
            t.split_kblocks = kr >> 7;
            t.split_rows = M;
            t.out_cols = N;
            t.partial = (float*)(ws + w.partial);
            rc = launch_moe_gemm_fp8w_256i(MODE_PLAIN, t, t256 * kt, s);
            if (rc != SGLK_OK) return rc;
            GenericGemmParams r{};
            r.partial = t.partial;
            r.ksplit = kt;
            r.split_rows = M;
            r.n_out = N;
            r.out = a->out;
            r.out_type = SGLK_OUT_BF16;
            r.out_stride = a->out_stride;
            r.bias = a->bias;
            return launch_splitk_reduce(r, s);
        }
        rc = launch_dense_tiles(M, 256, tile_info, num_tiles, ident, s);
        if (rc != SGLK_OK) return rc;
        MoeGemmParams t{};
        fill_tuned(t, a->x, a->x_stride, M, ident, a->w, a->w_scale, N, K, a->block_n, tile_info, num_tiles);
        t.n_tiles = N / 256;
        t.out = (uint16_t*)a->out;
        t.out_stride = a->out_stride;
        t.bias = a->bias;
        return launch_moe_gemm_fp8w_256i(MODE_PLAIN, t, t256, s);
    }
    // fp8, decode-size M: weight-streaming mid kernel (one "expert"), K cut into ranges until ~2 workgroups per CU exist
    if (a->wtype == SGLK_W_FP8_E4M3 && a->packed && !a->x_is_int8 && a->out_type == SGLK_OUT_BF16 && a->block_k == 128 &&
        a->block_n > 0 && a->block_n % 16 == 0 && a->x_stride % 8 == 0 && ((uintptr_t)a->x % 16) == 0 && a->out_stride % 4 == 0 &&
        ((uintptr_t)a->out % 8) == 0 && (!a->bias || ((uintptr_t)a->bias % 16) == 0) && !knobs().force_generic) {
        const int ks = (M < 1024 || dense_prefers_mid(M, N, a->wtype)) ? mid_dense_ksplit(M, N, K) : 0;   // from 1024 rows on only by the policy's choice
        if (ks >= 1) {
            const int mt = (int)ceil_div(M, kMidTileM);
            MoeGemmParams t{};   // no tile table: the kernel derives the row tiles from split_rows
            fill_tuned(t, a->x, a->x_stride, M, nullptr, a->w, a->w_scale, N, K, a->block_n, nullptr, nullptr);
            t.n_tiles = N / 128;
            t.out = (uint16_t*)a->out;
            t.out_stride = a->out_stride;
            t.bias = a->bias;
            t.ksplit = ks;
            t.split_kblocks = (K >> 7) / ks;
            t.split_rows = M;
            t.out_cols = N;
            t.partial = ks > 1 ? (float*)(ws + w.partial) : nullptr;
            rc = launch_moe_gemm_fp8w_mid(MODE_PLAIN, t, mt, s);
            if (rc != SGLK_OK || ks == 1) return rc;
            GenericGemmParams r{};
            r.partial = t.partial;
            r.ksplit = ks;
            r.split_rows = M;
            r.n_out = N;
            r.out = a->out;
            r.out_type = SGLK_OUT_BF16;
            r.out_stride = a->out_stride;
            r.bias = a->bias;
            return launch_splitk_reduce(r, s);
        }
    }
    // bf16 packed weights, decode-size M: the same weight-streaming scheme without a conversion (gemm_bf16_mid.hip)
    if (a->wtype == SGLK_W_BF16 && a->packed && !a->x_is_int8 && a->out_type == SGLK_OUT_BF16 && a->x_stride % 8 == 0 &&
        ((uintptr_t)a->x % 16) == 0 && a->out_stride % 4 == 0 && ((uintptr_t)a->out % 8) == 0 &&
        (!a->bias || ((uintptr_t)a->bias % 16) == 0) && ((uintptr_t)a->w % 4) == 0 && !knobs().force_generic) {
        // from 192 rows on only when the 256-row kernel below would have too few workgroups (or cannot take the shape)
        const bool tuned_can = N % 256 == 0 && K % 32 == 0 && a->out_stride % 8 == 0 && ((uintptr_t)a->out % 16) == 0;
        const int ks = (M < 192 || (!tuned_can && M < 1024) || dense_prefers_mid(M, N, a->wtype)) ? bf16_mid_ksplit(M, N, K) : 0;
        if (ks >= 1) {
            BmidParams q{};
            q.x = (const uint16_t*)a->x;
            q.x_stride = a->x_stride;
            q.w = (const uint8_t*)a->w;
            q.bias = a->bias;
            q.out = (uint16_t*)a->out;
            q.out_stride = a->out_stride;
            q.M = M; q.N = N; q.K = K;
            q.ksplit = ks;
            q.split_kblocks = (K >> 7) / ks;
            q.partial = ks > 1 ? (float*)(ws + w.partial) : nullptr;
            rc = launch_gemm_bf16_mid(q, s);
            if (rc != SGLK_OK || ks == 1) return rc;
            GenericGemmParams r{};
            r.partial = q.partial;
            r.ksplit = ks;
            r.split_rows = M;
            r.n_out = N;
            r.out = a->out;
            r.out_type = SGLK_OUT_BF16;
            r.out_stride = a->out_stride;
            r.bias = a->bias;
            return launch_splitk_reduce(r, s);
        }
    }
    // bf16 with the reference's packed (VNNI-2) weights on the bf16 matrix cores: large M
    if (a->wtype == SGLK_W_BF16 && a->packed && !a->x_is_int8 && M >= 192 && N % 256 == 0 && K % 32 == 0 && K >= 128 &&
        a->out_type == SGLK_OUT_BF16 && a->out_stride % 8 == 0 && ((uintptr_t)a->out % 16) == 0 && a->x_stride % 8 == 0 &&
        ((uintptr_t)a->x % 16) == 0 && (int64_t)M * a->x_stride * 2 < (1ll << 32) && (int64_t)N * K * 2 < (1ll << 32) &&
        !knobs().force_generic) {
        Bf16GemmParams q{};
        q.x = (const uint16_t*)a->x;
        q.x_stride = a->x_stride * 2;
        q.x_bytes = (int64_t)M * a->x_stride * 2;
        q.w = (const uint8_t*)a->w;
        q.w_bytes = (int64_t)N * K * 2;
        q.bias = a->bias;
        q.out = (uint16_t*)a->out;
        q.out_stride = a->out_stride;
        q.M = M;
        q.K = K;
        q.n_tiles = N / 256;
        const int kt = tuned_fp8_ksplit(M, N, K);         // the same policy as the fp8 tile kernel: long ranges, under-filled chip
        if (kt > 1 && w.partial && (K / kt) % 32 == 0 && K / kt >= 128) {
            q.K = K / kt;
            q.k_full = K;
            q.ksplit = kt;
            q.out_cols = N;
            q.partial = (float*)(ws + w.partial);
            q.bias = nullptr;
            rc = launch_gemm_bf16_256(MODE_PLAIN, q, (int)ceil_div(M, 256), s);
            if (rc != SGLK_OK) return rc;
            GenericGemmParams r{};
            r.partial = q.partial;
            r.ksplit = kt;
            r.split_rows = M;
            r.n_out = N;
            r.out = a->out;
            r.out_type = SGLK_OUT_BF16;
            r.out_stride = a->out_stride;
            r.bias = a->bias;
            return launch_splitk_reduce(r, s);
        }
        return launch_gemm_bf16_256(MODE_PLAIN, q, (int)ceil_div(M, 256), s);
    }
    // W8A8 at decode sizes and, while the 256-row kernel below would have at most 64 workgroups (or cannot run: M < 192), up to
    // SGLK_DENSE_MID_MAX rows: weight-streaming int8 kernel, exact int32 split-K partials (gemm_i8_mid.hip).  Same-box A/B
    // (tools/ab_i8_dense_mid.py, profiles/r03_ab_dense_129_1000.txt): 160 x 4096 x 4096 55 -> 21 us (the generic engine ran 129 ... 191
    // rows), 512 x 4096 x 4096 38 -> 29, 512 x 2048 x 6144 36 -> 25; 384 x 5120 x 2048 (40 workgroups) and 12288-wide layers from 192
    // rows on are faster on the tile kernel and stay there
    if (a->wtype == SGLK_W_INT8 && a->packed && !knobs().force_generic && !knobs().no_i8_mid) {
        const bool few = knobs().i8_dense_mid_wgs > 0 && M < knobs().dense_mid_max &&
                         (M < 192 || N % 256 != 0 || ceil_div(M, 256) * (int64_t)(N / 256) <= knobs().i8_dense_mid_wgs);
        const int ks = M <= 128 ? i8_mid_ksplit(M, N, K) : (few ? i8_mid_dense_ksplit(M, N, K) : 0);
        if (ks >= 1) {
            const int8_t* xq = (const int8_t*)a->x;
            int64_t xq_stride = a->x_stride;
            const float* xs = a->x_scale;
            if (quant_here) {
                rc = launch_quant_int8_rows((const uint16_t*)a->x, a->x_stride, (int8_t*)(ws + w.xq), K, (float*)(ws + w.xs),
                                            M, K, 1e-10f, s);
                if (rc != SGLK_OK) return rc;
                xq = (const int8_t*)(ws + w.xq);
                xq_stride = K;
                xs = (const float*)(ws + w.xs);
            }
            if (xq_stride % 16 == 0 && ((uintptr_t)xq % 16) == 0) {
                I8GemmParams q{};
                q.x = xq;
                q.x_stride = xq_stride;
                q.x_scale = xs;
                q.w = (const uint8_t*)a->w;
                q.w_bytes = (int64_t)N * K;
                q.w_scale = a->w_scale;
                q.scale_rows = N;
                q.bias = a->bias;
                q.out = (uint16_t*)a->out;
                q.out_stride = a->out_stride;
                q.out_type = a->out_type;
                q.M = M;
                q.N = N;
                q.K = K;
                q.ksplit = ks;
                q.split_kblocks = (K >> 7) / ks;
                q.partial_i32 = ks > 1 ? (int32_t*)(ws + w.partial) : nullptr;
                return launch_gemm_i8_mid_plain(q, s);
            }
        }
    }
    // W8A8 on the int8 matrix cores (exact int32 accumulation): packed int8 weights, large M
    if (a->wtype == SGLK_W_INT8 && a->packed && M >= 192 && N % 256 == 0 && K % 64 == 0 && K >= 256 && a->out_type == SGLK_OUT_BF16 &&
        a->out_stride % 8 == 0 && ((uintptr_t)a->out % 16) == 0 && (int64_t)N * K < (1ll << 32) &&
        !knobs().force_generic) {
        const int8_t* xq = (const int8_t*)a->x;
        int64_t xq_stride = a->x_stride;
        const float* xs = a->x_scale;
        if (quant_here) {
            rc = launch_quant_int8_rows((const uint16_t*)a->x, a->x_stride, (int8_t*)(ws + w.xq), K, (float*)(ws + w.xs),
                                        M, K, 1e-10f, s);
            if (rc != SGLK_OK) return rc;
            xq = (const int8_t*)(ws + w.xq);
            xq_stride = K;
            xs = (const float*)(ws + w.xs);
        }
        if (xq_stride % 16 == 0 && ((uintptr_t)xq % 16) == 0 && (int64_t)M * xq_stride < (1ll << 32)) {
            I8GemmParams q{};
            q.x = xq;
            q.x_stride = xq_stride;
            q.x_bytes = (int64_t)M * xq_stride;
            q.x_scale = xs;
            q.w = (const uint8_t*)a->w;
            q.w_bytes = (int64_t)N * K;
            q.w_scale = a->w_scale;
            q.scale_rows = N;
            q.bias = a->bias;
            q.out = (uint16_t*)a->out;
            q.out_stride = a->out_stride;
            q.M = M;
            q.K = K;
            q.n_tiles = N / 256;
            // under-filled launch, long ranges: exact int32 partials per K range.  The int8 tiles are twice as fast, so the partials'
            // round trip only pays below an eighth of the chip (1024 x 2048 x 6144: 0.066 -> 0.051 ms; 2048 x 4096 x 4096 would lose)
            const int kt = tuned_fp8_ksplit(M, N, K, 8);
            // enough rows to fill the chip: 128-row tiles, two workgroups per CU (moe_gemm_fp8w_s128.hip, MODE_PLAIN, terms = 0); the
            // same exact int32 sums and the same separately rounded epilogue, bit for bit
            if (kt == 1 && dense_s128_shape_ok(M, N, K)) {
                const int t128 = (int)ceil_div(M, 128);
                A8GemmParams p8{};
                p8.x = (const uint8_t*)xq;
                p8.x_stride = xq_stride;
                p8.x_bytes = (int64_t)M * xq_stride;
                p8.x_scale_f32 = xs;
                p8.topk = 1;
                p8.w = (const uint8_t*)a->w;
                p8.w_expert_stride = (int64_t)N * K;
                p8.w_scale = a->w_scale;
                p8.scale_rows = N;
                p8.C = K;
                p8.dense_rows = M;
                p8.n_tiles = N / 256;
                p8.out = a->out;
                p8.out_stride = a->out_stride;
                p8.bias = a->bias;
                p8.max_mtiles = t128;
                return launch_moe_gemm_fp8w_s128(MODE_PLAIN, p8, t128, s, 0);
            }
            if (kt > 1 && w.partial) {
                q.K = K / kt;
                q.N = N;
                q.ksplit = kt;
                q.split_kblocks = (K / kt) >> 7;
                q.partial_i32 = (int32_t*)(ws + w.partial);
                q.out_type = SGLK_OUT_BF16;
                rc = launch_gemm_i8_256(MODE_PLAIN, q, (int)ceil_div(M, 256), s);
                if (rc != SGLK_OK) return rc;
                return launch_i8_splitk_reduce(q, s);
            }
            return launch_gemm_i8_256(MODE_PLAIN, q, (int)ceil_div(M, 256), s);
        }
    }
    const int tiles = (int)ceil_div(M, kGenericTileM);
    GenericGemmParams g{};
    g.dense_rows = M;             // m-tiles derived from M inside the kernel: no table launch
    g.x = a->x;
    g.x_type = a->x_is_int8 ? SGLK_W_INT8 : SGLK_W_BF16;
    g.x_stride = a->x_stride;
    g.x_row_scale = a->x_scale;
    if (quant_here) {
        rc = launch_quant_int8_rows((const uint16_t*)a->x, a->x_stride, (int8_t*)(ws + w.xq), K, (float*)(ws + w.xs),
                                    M, K, 1e-10f, s);
        if (rc != SGLK_OK) return rc;
        g.x = ws + w.xq;
        g.x_type = SGLK_W_INT8;
        g.x_stride = K;
        g.x_row_scale = (const float*)(ws + w.xs);
    }
    g.topk = 1;
    g.gather = GG_GATHER_NONE;
    g.tile_info = tile_info;
    g.num_tiles = num_tiles;
    g.n_tiles = (int)ceil_div(N, 64);
    fill_weight(g, a->w, a->w_scale, a->wtype, a->packed, N, K, a->block_n);
    g.n_out = N;
    g.out = a->out;
    g.out_type = a->out_type;
    g.out_stride = a->out_stride;
    g.bias = a->bias;
    g.ksplit = generic_ksplit(M, N, K);
    if (g.ksplit > 1) {
        const int stages = (int)ceil_div(K, 64);
        int per = (int)ceil_div(stages, g.ksplit);
        per += per & 1;                                   // whole fp8 K blocks
        g.split_stages = per;
        g.ksplit = (int)ceil_div(stages, per);
        g.split_rows = M;
        g.partial = (float*)(ws + w.partial);
    }
    return launch_gemm_generic(GG_PLAIN, g, tiles, s);
}

extern "C" size_t sglk_mxfp4_workspace_bytes(int32_t M, int32_t N, int32_t K) {
    if (M < 0 || N <= 0 || K <= 0) return 0;
    const size_t dense = plan_dense(M, N, K, false, false).total, native = mxfp4_native_workspace_bytes(M, N, K);
    return dense > native ? dense : native;
}

extern "C" int sglk_mxfp4_scaled_mm(const void* x, int64_t x_stride, const void* wq, const void* scales, int32_t scale_packed,
                                    const float* bias, void* out, int64_t out_stride, int32_t M, int32_t N, int32_t K,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    SGLK_REQUIRE(M >= 0 && N > 0 && K > 0, SGLK_ERR_INVALID, "mxfp4_scaled_mm: bad sizes M=%d N=%d K=%d", M, N, K);
    SGLK_REQUIRE(K % 32 == 0, SGLK_ERR_SHAPE, "mxfp4_scaled_mm: K (%d) must be a multiple of the 32-wide scale block", K);
    SGLK_REQUIRE(!scale_packed || N % 32 == 0, SGLK_ERR_SHAPE, "mxfp4_scaled_mm: packed scales need N (%d) %% 32 == 0", N);
    SGLK_REQUIRE(wq && scales && workspace, SGLK_ERR_INVALID, "mxfp4_scaled_mm: null pointer");
    SGLK_REQUIRE(M == 0 || (x && out), SGLK_ERR_INVALID, "mxfp4_scaled_mm: null pointer");
    SGLK_REQUIRE(x_stride >= K && out_stride >= N, SGLK_ERR_INVALID, "mxfp4_scaled_mm: row stride too small");
    SGLK_REQUIRE(((uintptr_t)wq % 4) == 0, SGLK_ERR_INVALID, "mxfp4_scaled_mm: weights must be 4-byte aligned");
    const DenseWs w = plan_dense(M, N, K, false, false);
    SGLK_REQUIRE(workspace_bytes >= w.total, SGLK_ERR_WORKSPACE, "mxfp4_scaled_mm: workspace %zu < required %zu",
                 workspace_bytes, w.total);
    if (M == 0) return SGLK_OK;
    hipStream_t s = (hipStream_t)stream;
    // M >= 64 and tileable shapes: the fp4 weights as stored on the block-scaled matrix cores (gemm_mxfp4.hip); a caller that
    // sized the workspace with sglk_scaled_mm_workspace_bytes only (v1) stays on the bf16 expansion below
    if (mxfp4_native_ok(M, N, K, x, x_stride, wq, out, out_stride) && workspace_bytes >= mxfp4_native_workspace_bytes(M, N, K))
        return launch_gemm_mxfp4_native(x, x_stride, wq, scales, scale_packed, bias, out, out_stride, M, N, K, workspace, s);
    unsigned char* ws = (unsigned char*)workspace;
    int4* tile_info = (int4*)(ws + w.tile_info);
    int* num_tiles = (int*)(ws + w.num_tiles);
    const int tiles = (int)ceil_div(M, kGenericTileM);
    int rc = SGLK_OK;
    (void)rc;
    GenericGemmParams g{};
    g.dense_rows = M;             // m-tiles derived from M inside the kernel: no table launch
    g.x = x;
    g.x_type = SGLK_W_BF16;
    g.x_stride = x_stride;
    g.topk = 1;
    g.gather = GG_GATHER_NONE;
    g.tile_info = tile_info;
    g.num_tiles = num_tiles;
    g.n_tiles = (int)ceil_div(N, 64);
    g.w = wq;
    g.w_type = SGLK_W_MXFP4;
    g.packed = 0;
    g.C = K;
    g.w_scale = (const float*)scales;   // E8M0 bytes; the kernel reads them as such
    g.block_n = scale_packed ? 1 : 0;   // their layout
    g.scale_rows = N;
    g.scale_cols = K / 32;
    g.n_out = N;
    g.out = out;
    g.out_type = SGLK_OUT_BF16;
    g.out_stride = out_stride;
    g.bias = bias;
    g.ksplit = generic_ksplit(M, N, K);
    if (g.ksplit > 1) {
        const int stages = (int)ceil_div(K, 64);
        int per = (int)ceil_div(stages, g.ksplit);
        per += per & 1;
        g.split_stages = per;
        g.ksplit = (int)ceil_div(stages, per);
        g.split_rows = M;
        g.partial = (float*)(ws + w.partial);
    }
    return launch_gemm_generic(GG_PLAIN, g, tiles, s);
}
