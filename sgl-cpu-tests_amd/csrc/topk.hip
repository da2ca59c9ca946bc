// DeepSeek-style grouped expert routing: grouped_topk (softmax) and biased_grouped_topk (sigmoid + correction bias).
//
// Oracles: /root/reference/test_grouped_topk.py:9-39 and /root/reference/test_biased_grouped_topk.py:9-47:
//   scores = softmax(gating) | sigmoid(gating);  choice = scores (| + bias)
//   group score = max of the group's scores | sum of the group's two largest choices
//   keep the topk_group best groups, take the topk best experts among them by `choice`,
//   weights = scores at those ids, optionally renormalised to sum 1.
// One wave per token, everything in fp32.  Tie rule (the reference's torch.topk leaves it open): larger value
// first, equal values -> lower index first; experts of non-selected groups are only taken when the selected groups
// hold fewer than topk experts, and then carry weight 0 (softmax variant, matching masked_fill(0.0)).
// Softmax variant: softmax is monotone, so groups and experts are RANKED BY THE LOGITS (exact comparisons of the inputs)
// and only the returned weights go through exp(): the ids do not depend on anybody's exp rounding and equal this repo's
// oracle bit for bit (oracle/routing.py ranks equal fp32 scores by logit, then index -- a refinement of the open tie rule).
// Sigmoid + bias variant: the ranking key sigmoid(x) + bias is a rounded quantity; ids equal the oracle's except on
// genuine near-ties of that key (asserted as such in tests/test_rows_topk_gpu.py).  Output order: descending key.
#include "moe_align_small.h"
#include "moe_internal.h"

namespace sglk {

constexpr int kTopkMaxE = 1024;   // experts per token: 16 per lane
// router + align in one workgroup: ONE token per wave.  More tokens per wave lose: the sixteen waves of the single workgroup
// all reduce through the one CU's cross-lane (ds_bpermute) path -- measured 14 us per extra round of 16 tokens (M = 64: 196 vs
// 140 us for the whole block), while the stand-alone router spreads its tokens over many CUs.
constexpr int kRouteAlignMaxTokens = 16;
constexpr int kRouteAlignMaxSlots = 1024;
constexpr int kPerLane = kTopkMaxE / 64;

// Cross-lane reductions on the DPP path of the vector ALU, not through ds_bpermute: four steps inside the rows of 16 lanes
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: after them every lane of a row holds the row's result), then the four rows'
// results are read as scalars.  A routed token needs ~25 of these one after the other; as bpermute chains (a trip through the LDS
// unit per step, six steps each) they were most of the kernel's 52 us at 16384 tokens.
template <int CTRL>
SGLK_DEV float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
SGLK_DEV int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
SGLK_DEV float lane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

SGLK_DEV float wave_max(float v) {
    v = fmaxf(v, dpp_f<kDppXor1>(v));
    v = fmaxf(v, dpp_f<kDppXor2>(v));
    v = fmaxf(v, dpp_f<kDppHalfMirror>(v));
    v = fmaxf(v, dpp_f<kDppMirror>(v));
    return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}
SGLK_DEV float wave_sum(float v) {
    v += dpp_f<kDppXor1>(v);
    v += dpp_f<kDppXor2>(v);
    v += dpp_f<kDppHalfMirror>(v);
    v += dpp_f<kDppMirror>(v);
    return ((lane_f(v, 0) + lane_f(v, 16)) + lane_f(v, 32)) + lane_f(v, 48);
}
// arg-max over the wave of (value, index): larger value wins, ties -> lower index.  (value desc, index asc) is a total order on
// NaN-free input, so the winner does not depend on the shape of the reduction tree.  The result is wave-uniform.
template <int CTRL>
SGLK_DEV void argmax_step(float& v, int& i) {
    const float ov = dpp_f<CTRL>(v);
    const int oi = dpp_i<CTRL>(i);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
SGLK_DEV void wave_argmax(float& v, int& i) {
    argmax_step<kDppXor1>(v, i);
    argmax_step<kDppXor2>(v, i);
    argmax_step<kDppHalfMirror>(v, i);
    argmax_step<kDppMirror>(v, i);
    float bv = lane_f(v, 0);
    int bi = __builtin_amdgcn_readlane(i, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const float ov = lane_f(v, r);
        const int oi = __builtin_amdgcn_readlane(i, r);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    v = bv;
    i = bi;
}

// The picks run on ONE 64-bit key per candidate: (order-preserving image of the fp32 value) << 32 | ~index, so that the unsigned
// maximum is "larger value first, equal values -> lower index first" and a step of the reduction is two DPP moves, one 64-bit
// compare and two selects.  Key 0 = not a candidate (every real value, -inf included, maps above it).  The four rows are combined
// with row_bcast15 / row_bcast31 (the wave's maximum ends up in row 3) and read from lane 63: wave-uniform.
typedef unsigned long long u64;
SGLK_DEV u64 topk_key(float v, int index) {
    const unsigned b = __float_as_uint(v + 0.f);     // -0 -> +0: the two compare equal as floats and must tie here too
    const unsigned ord = b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
    return ((u64)ord << 32) | (unsigned)(~index);
}
SGLK_DEV int topk_key_index(u64 k) { return (int)(~(unsigned)k); }
template <int CTRL, int ROW_MASK>
SGLK_DEV u64 key_max_step(u64 k) {      // lanes of rows outside ROW_MASK read their own key back
    const int lo = (int)(unsigned)k, hi = (int)(unsigned)(k >> 32);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const u64 o = ((u64)ohi << 32) | olo;
    return o > k ? o : k;
}
SGLK_DEV u64 wave_max_key(u64 k) {
    k = key_max_step<kDppXor1, 0xf>(k);
    k = key_max_step<kDppXor2, 0xf>(k);
    k = key_max_step<kDppHalfMirror, 0xf>(k);
    k = key_max_step<kDppMirror, 0xf>(k);
    k = key_max_step<0x142, 0xa>(k);     // row_bcast15: rows 1, 3 take in rows 0, 2
    k = key_max_step<0x143, 0xc>(k);     // row_bcast31: rows 2, 3 take in rows 0 + 1
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), 63);
    return ((u64)hi << 32) | lo;
}

template <int GT>   // gating element type: 0 bf16, 1 f16, 2 f32
SGLK_DEV float ld_gate(const void* p, int64_t idx) {
    if (GT == 2) return reinterpret_cast<const float*>(p)[idx];
    const unsigned short b = reinterpret_cast<const unsigned short*>(p)[idx];
    if (GT == 1) return (float)__builtin_bit_cast(_Float16, b);
    return bf16_bits_to_f32(b);
}

// Routing of ONE token by one wave.  PL = experts per lane (E <= 64 * PL).  choice_l [64 * PL floats] and gscore_l are this
// wave's LDS scratch.  Lanes < topk return their pick in (my_w, my_id), `wsum` is the sum of the picked scores on all lanes.
template <int GT, bool BIASED, int PL>
SGLK_DEV void route_one_token(const void* __restrict__ gating, int64_t g_stride, const void* __restrict__ bias, int m, int E,
                              int topk, int G, int topk_group, float* choice_l, int lane, float& my_w, int& my_id, float& wsum) {
    const int per_group = E / G;
    // ---- scores ---------------------------------------------------------------------------------------------------
    float score[PL], choice[PL];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < PL; ++j) {
        const int e = j * 64 + lane;
        score[j] = e < E ? ld_gate<GT>(gating, (int64_t)m * g_stride + e) : -INFINITY;
        choice[j] = score[j];   // softmax variant: the ranking key is the logit itself
        mx = fmaxf(mx, score[j]);
    }
    if (!BIASED) {
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < PL; ++j) {
            score[j] = (j * 64 + lane < E) ? expf(score[j] - mx) : 0.f;
            sum += score[j];
        }
        sum = wave_sum(sum);
#pragma unroll
        for (int j = 0; j < PL; ++j) score[j] /= sum;
    } else {
#pragma unroll
        for (int j = 0; j < PL; ++j) {
            const int e = j * 64 + lane;
            if (e < E) {
                score[j] = 1.0f / (1.0f + expf(-score[j]));
                choice[j] = score[j] + ld_gate<GT>(bias, e);
            } else {
                score[j] = 0.f;
                choice[j] = -INFINITY;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < PL; ++j)
        if (j * 64 + lane < E) choice_l[j * 64 + lane] = choice[j];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- group scores, one lane per group (G <= 64) ------------------------------------------------------------------
    // (one group -- plain top-k routing, Qwen3 -- is selected whatever its score: nothing to rank.  The general path below had lane 0
    // walk all E experts through LDS, one dependent read after the other.)
    float gs = -INFINITY;
    if (G > 1 && lane < G) {
        float a = -INFINITY, b = -INFINITY;   // two largest of the group
        for (int i = 0; i < per_group; ++i) {
            const float v = choice_l[lane * per_group + i];
            if (v > a) { b = a; a = v; } else if (v > b) { b = v; }
        }
        gs = BIASED ? a + b : a;
    }
    // ---- the topk_group best groups -> bitmask ----------------------------------------------------------------------------
    unsigned long long gmask = G == 1 ? 1ull : 0ull;
    for (int r = 0; G > 1 && r < topk_group; ++r) {
        // a group whose score is -inf is still a candidate (its key is above 0); taken groups and padding lanes are not
        const u64 k = wave_max_key((lane < G && !((gmask >> lane) & 1ull)) ? topk_key(gs, lane) : 0ull);
        if (k != 0ull) gmask |= 1ull << topk_key_index(k);
    }
    // ---- topk experts among the selected groups --------------------------------------------------------------------------
    float masked[PL];
#pragma unroll
    for (int j = 0; j < PL; ++j) {
        const int e = j * 64 + lane;
        const bool in = e < E && ((gmask >> (e / per_group)) & 1ull);
        masked[j] = in ? choice[j] : -INFINITY;
    }
    u64 key[PL];
#pragma unroll
    for (int j = 0; j < PL; ++j) key[j] = masked[j] > -INFINITY ? topk_key(masked[j], j * 64 + lane) : 0ull;   // e >= E is -inf already
    wsum = 0.f;
    my_w = 0.f;
    my_id = 0;
    for (int r = 0; r < topk; ++r) {
        u64 kb = key[0];
#pragma unroll
        for (int j = 1; j < PL; ++j) kb = key[j] > kb ? key[j] : kb;
        kb = wave_max_key(kb);
        // second chance: nothing left in the selected groups -> lowest-index expert not taken yet (weight 0 / raw score)
        int i = kb != 0ull ? topk_key_index(kb) : (1 << 20);
        float w_sel = 0.f;
        if (i >= E) {   // all remaining candidates are -inf: take the lowest index still marked "not taken"
            int cand = 1 << 20;
#pragma unroll
            for (int j = 0; j < PL; ++j) {
                const int e = j * 64 + lane;
                if (e < E && masked[j] == -INFINITY && choice[j] != INFINITY && e < cand) cand = e;   // choice==INF marks taken
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int oc = __shfl_xor(cand, o); cand = oc < cand ? oc : cand; }
            i = cand < E ? cand : 0;
        }
        // owner lane retires the pick and supplies its (unbiased) score; i is wave-uniform: the score is read from the owner's lane
#pragma unroll
        for (int j = 0; j < PL; ++j) {
            if (j * 64 + lane == i) {
                const bool was_selected_group = masked[j] != -INFINITY;
                w_sel = (BIASED || was_selected_group) ? score[j] : 0.f;
                masked[j] = -INFINITY;
                key[j] = 0ull;
                choice[j] = INFINITY;   // taken
            }
        }
        w_sel = lane_f(w_sel, __builtin_amdgcn_readfirstlane(i) & 63);
        wsum += w_sel;
        if (lane == r) { my_w = w_sel; my_id = i; }
    }
}

template <int GT, bool BIASED, int PL>   // PL = experts per lane (E <= 64 * PL): 2 / 4 for the usual 128 / 256 experts, 16 up to 1024
__global__ __launch_bounds__(256) void grouped_topk_kernel(const void* __restrict__ gating, int64_t g_stride,
                                                           const void* __restrict__ bias, float* __restrict__ out_w,
                                                           int* __restrict__ out_ids, int M, int E, int topk,
                                                           int renormalize, int G, int topk_group) {
    __shared__ float s_choice[4][64 * PL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x * 4 + wave;
    if (m >= M) return;   // whole wave; no block-level barrier is used below
    float my_w, wsum;
    int my_id;
    route_one_token<GT, BIASED, PL>(gating, g_stride, bias, m, E, topk, G, topk_group, s_choice[wave], lane, my_w, my_id, wsum);
    if (lane < topk) {
        out_ids[(int64_t)m * topk + lane] = my_id;
        out_w[(int64_t)m * topk + lane] = renormalize ? my_w / wsum : my_w;
    }
}


// ---- router + align in ONE launch (no reference counterpart: the reference harness calls
//      grouped_topk_cpu and fused_experts_cpu one after the other, /root/reference/test_moe.py:57-92) -------------------------
// One workgroup of 16 waves: wave w routes tokens w, w + 16, ... exactly as grouped_topk_kernel does (same device function:
// ids and weights are bit-identical to the stand-alone operator), keeps the ids in LDS, and after one barrier the whole
// workgroup runs the small stable counting sort + tile table on them (moe_align_small.h).  For decode batches of up to 16
// tokens: saves the launch boundary between the two kernels and the global round trip of the ids (3.6 us of 47.6 at M = 1).
template <int GT, bool BIASED>
__global__ __launch_bounds__(1024) void route_align_kernel(const void* __restrict__ gating, int64_t g_stride,
                                                           const void* __restrict__ bias, float* __restrict__ out_w,
                                                           int* __restrict__ out_ids, int M, int E, int topk, int renormalize,
                                                           int G, int topk_group, int nbits, int tile_m, int max_tiles,
                                                           int* __restrict__ sorted_slot, int* __restrict__ expert_off,
                                                           int* __restrict__ tile_info, int* __restrict__ num_tiles) {
    __shared__ float s_choice[16][kSmallMaxE];
    __shared__ int s_ids[kRouteAlignMaxSlots];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int m = wave; m < M; m += 16) {
        float my_w, wsum;
        int my_id;
        route_one_token<GT, BIASED, kSmallMaxE / 64>(gating, g_stride, bias, m, E, topk, G, topk_group, s_choice[wave], lane, my_w,
                                                     my_id, wsum);
        if (lane < topk) {
            out_ids[(int64_t)m * topk + lane] = my_id;
            out_w[(int64_t)m * topk + lane] = renormalize ? my_w / wsum : my_w;
            s_ids[m * topk + lane] = my_id;
        }
    }
    __syncthreads();
    moe_align_small_body(s_ids, M * topk, E, nbits, tile_m, max_tiles, sorted_slot, expert_off, tile_info, num_tiles);
}

}  // namespace sglk

using namespace sglk;

namespace sglk {
bool route_align_ok(int M, int E, int topk) { return M >= 1 && M <= kRouteAlignMaxTokens && E <= kSmallMaxE && M * topk <= kRouteAlignMaxSlots; }

int launch_route_align(const void* gating, int64_t gating_stride, int gating_type, const void* bias, float* topk_weights,
                       int32_t* topk_ids, int M, int E, int topk, int renormalize, int G, int topk_group, int tile_m,
                       int32_t* sorted_slot, int32_t* expert_off, int32_t* tile_info, int32_t* num_tiles, hipStream_t s) {
    if (!route_align_ok(M, E, topk)) SGLK_FAIL(SGLK_ERR_SHAPE, "route_align: M=%d E=%d topk=%d outside the one-workgroup range", M, E, topk);
    int nbits = 0;
    while ((1 << nbits) < E) ++nbits;
    const int max_tiles = sglk_moe_max_tiles(M, E, topk, tile_m);
#define RA_LAUNCH(GT, B)                                                                                                \
    hipLaunchKernelGGL((route_align_kernel<GT, B>), dim3(1), dim3(1024), 0, s, gating, gating_stride, bias, topk_weights, topk_ids, \
                       M, E, topk, renormalize, G, topk_group, nbits, tile_m, max_tiles, sorted_slot, expert_off, tile_info, num_tiles)
    if (bias) {
        if (gating_type == 0) RA_LAUNCH(0, true);
        else if (gating_type == 1) RA_LAUNCH(1, true);
        else RA_LAUNCH(2, true);
    } else {
        if (gating_type == 0) RA_LAUNCH(0, false);
        else if (gating_type == 1) RA_LAUNCH(1, false);
        else RA_LAUNCH(2, false);
    }
#undef RA_LAUNCH
    SGLK_CHECK_LAUNCH("route_align");
    return SGLK_OK;
}
}  // namespace sglk

extern "C" int sglk_grouped_topk(const void* gating, int64_t gating_stride, int32_t gating_type, const void* bias,
                                 float* topk_weights, int32_t* topk_ids, int32_t M, int32_t E, int32_t topk,
                                 int32_t renormalize, int32_t num_expert_group, int32_t topk_group, void* stream) {
    SGLK_REQUIRE(M >= 0 && E > 0 && topk > 0, SGLK_ERR_INVALID, "grouped_topk: bad sizes M=%d E=%d topk=%d", M, E, topk);
    SGLK_REQUIRE(E <= kTopkMaxE, SGLK_ERR_SHAPE, "grouped_topk: at most %d experts (got %d)", kTopkMaxE, E);
    SGLK_REQUIRE(topk <= 64 && topk <= E, SGLK_ERR_SHAPE, "grouped_topk: topk must be <= min(64, E) (got %d)", topk);
    SGLK_REQUIRE(num_expert_group > 0 && num_expert_group <= 64 && E % num_expert_group == 0, SGLK_ERR_SHAPE,
                 "grouped_topk: num_expert_group (%d) must divide E (%d) and be <= 64", num_expert_group, E);
    SGLK_REQUIRE(topk_group > 0 && topk_group <= num_expert_group, SGLK_ERR_SHAPE,
                 "grouped_topk: topk_group (%d) must be in [1, num_expert_group]", topk_group);
    SGLK_REQUIRE(gating_type >= 0 && gating_type <= 2, SGLK_ERR_INVALID, "grouped_topk: bad gating_type");
    SGLK_REQUIRE(M == 0 || (gating && topk_weights && topk_ids), SGLK_ERR_INVALID, "grouped_topk: null pointer");
    SGLK_REQUIRE(gating_stride >= E, SGLK_ERR_INVALID, "grouped_topk: gating stride < E");
    if (M == 0) return SGLK_OK;
    const dim3 grid((unsigned)ceil_div(M, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    // every per-lane loop of the router runs PL times: sixteen for 128 experts did eight times the work (M = 16384: 88 us)
#define TOPK_LAUNCH_PL(GT, B, PL)                                                                                        \
    hipLaunchKernelGGL((grouped_topk_kernel<GT, B, PL>), grid, block, 0, s, gating, gating_stride, bias, topk_weights, \
                       topk_ids, M, E, topk, renormalize, num_expert_group, topk_group)
#define TOPK_LAUNCH(GT, B)                                  \
    do {                                                    \
        if (E <= 128) TOPK_LAUNCH_PL(GT, B, 2);             \
        else if (E <= 256) TOPK_LAUNCH_PL(GT, B, 4);        \
        else TOPK_LAUNCH_PL(GT, B, kPerLane);               \
    } while (0)
    if (bias) {
        if (gating_type == 0) TOPK_LAUNCH(0, true);
        else if (gating_type == 1) TOPK_LAUNCH(1, true);
        else TOPK_LAUNCH(2, true);
    } else {
        if (gating_type == 0) TOPK_LAUNCH(0, false);
        else if (gating_type == 1) TOPK_LAUNCH(1, false);
        else TOPK_LAUNCH(2, false);
    }
#undef TOPK_LAUNCH
    SGLK_CHECK_LAUNCH("grouped_topk");
    return SGLK_OK;
}
