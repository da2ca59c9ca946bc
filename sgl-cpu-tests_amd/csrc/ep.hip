// Expert-parallel dispatch glue as kernels (no reference counterpart: the reference pins only the local contract EP needs,
// topk_ids == -1 for experts that are not resident, /root/reference/test_moe_offloading_cpu.py:12-15,62-68).
//
//   ep_plan : for every destination rank d, the tokens that route at least one slot to an expert of d, in ascending token
//             order (a stable compaction, so the exchange is deterministic): counts[d] and pos[m][d] = index of token m
//             inside the segment for d (-1: not sent).  One launch of ceil(M / 1024) workgroups.
//   ep_pack : one payload row per (token, destination): [K bf16 | topk local ids i32 | topk routing weights f32], ids
//             rewritten to the destination's numbering and -1 elsewhere -- rows and routing travel in ONE all-to-all.
//             Segment of destination d starts at seg_start[d] rows (exact mode: exclusive sum of counts; capacity mode:
//             d * capacity, and the unused tail of every segment gets ids = -1 so the receiver's experts skip it).
#include "sglk_common.h"

namespace sglk {

constexpr int kEpMaxRanks = 16;

// Workgroup b plans tokens [1024 b, 1024 b + 1024).  It needs the number of earlier tokens per destination; instead of a
// chained scan over workgroups it simply RE-COUNTS them (all 1024 threads over the ids of tokens [0, 1024 b): at most 512 KB
// from L2 at M = 16384): one launch, no inter-workgroup synchronisation, deterministic.
// destination of an expert id without an integer division (8 per token, 16 tokens per thread in the last workgroup: they were most
// of the kernel's 158 us at M = 16384): (e + 0.5) / epr is at least 0.5 / epr away from an integer, far more than fp32 rounding
SGLK_DEV unsigned ep_dest_bit(int e, int E, float inv_epr) {
    return (unsigned)e < (unsigned)E ? 1u << (int)(((float)e + 0.5f) * inv_epr) : 0u;
}

template <bool VEC4>   // VEC4: topk % 4 == 0 and 16-byte aligned ids -> one int4 per four slots
__global__ __launch_bounds__(1024) void ep_plan_kernel(const int* __restrict__ ids, int M, int topk, int E, int G, float inv_epr,
                                                       int capacity, int* __restrict__ counts, int* __restrict__ seg_start,
                                                       int* __restrict__ pos, int* __restrict__ overflow) {
    __shared__ int wave_tot[16][kEpMaxRanks];
    __shared__ int base_s[kEpMaxRanks];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int first = blockIdx.x * 1024;
    auto mask_of = [&](int m) __attribute__((always_inline)) {
        unsigned mask = 0;
        if (VEC4) {
            const int4* row = reinterpret_cast<const int4*>(ids + (int64_t)m * topk);
            for (int j = 0; j < (topk >> 2); ++j) {
                const int4 v = row[j];
                mask |= ep_dest_bit(v.x, E, inv_epr) | ep_dest_bit(v.y, E, inv_epr) | ep_dest_bit(v.z, E, inv_epr) |
                        ep_dest_bit(v.w, E, inv_epr);
            }
        } else {
            for (int j = 0; j < topk; ++j) mask |= ep_dest_bit(ids[(int64_t)m * topk + j], E, inv_epr);
        }
        return mask;
    };
    // ---- tokens before this workgroup's range, per destination ------------------------------------------------------------
    int cnt[kEpMaxRanks];
#pragma unroll
    for (int d = 0; d < kEpMaxRanks; ++d) cnt[d] = 0;
#pragma unroll 4
    for (int m = tid; m < first; m += 1024) {
        const unsigned mask = mask_of(m);
#pragma unroll
        for (int d = 0; d < kEpMaxRanks; ++d) cnt[d] += (mask >> d) & 1u;
    }
#pragma unroll
    for (int d = 0; d < kEpMaxRanks; ++d) {
        int v = cnt[d];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) wave_tot[wave][d] = v;
    }
    __syncthreads();
    if (tid < kEpMaxRanks) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += wave_tot[w][tid];
        base_s[tid] = t;
    }
    __syncthreads();
    // ---- own token: exclusive scan over the workgroup's 1024 tokens, per destination -----------------------------------------
    const int m = first + tid;
    const unsigned mask = m < M ? mask_of(m) : 0u;
    int excl[kEpMaxRanks];
#pragma unroll
    for (int d = 0; d < kEpMaxRanks; ++d) {
        const int mine = (mask >> d) & 1u;
        int v = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(v, o);
            if (lane >= o) v += u;
        }
        excl[d] = v - mine;
        if (lane == 63) wave_tot[wave][d] = v;   // safe: the reads of the first use finished before the second barrier above
    }
    __syncthreads();
    const bool last = blockIdx.x == gridDim.x - 1;
    int run = 0;   // last workgroup, thread 0: exclusive sum of the totals = segment starts of the exact mode
#pragma unroll
    for (int d = 0; d < kEpMaxRanks; ++d) {
        if (d < G) {
            int before = 0, total = 0;
            for (int w = 0; w < 16; ++w) {
                const int t = wave_tot[w][d];
                before += w < wave ? t : 0;
                total += t;
            }
            const int b0 = base_s[d];
            if (m < M) {
                int p = -1;
                if ((mask >> d) & 1u) {
                    p = b0 + before + excl[d];
                    if (capacity > 0 && p >= capacity) p = -1;   // dropped (flagged in *overflow by the last workgroup)
                }
                pos[(int64_t)m * G + d] = p;
            }
            if (last && tid == 0) {
                const int all = b0 + total;
                counts[d] = all;
                seg_start[d] = capacity > 0 ? d * capacity : run;
                run += all;
                if (capacity > 0 && all > capacity) atomicOr(overflow, 1 << d);
            }
        }
    }
    if (last && tid == 0) seg_start[G] = capacity > 0 ? G * capacity : run;
}

// one wave per token: copy its row + routing to every destination segment it belongs to
__global__ __launch_bounds__(256) void ep_pack_kernel(const uint16_t* __restrict__ hidden, int64_t hidden_stride,
                                                      const int* __restrict__ ids, const float* __restrict__ tw,
                                                      const int* __restrict__ pos, const int* __restrict__ seg_start,
                                                      unsigned char* __restrict__ payload, int64_t row_bytes, int M, int K,
                                                      int topk, int G, int epr) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const uint16_t* src = hidden + (int64_t)m * hidden_stride;
    for (int d = 0; d < G; ++d) {
        const int p = pos[(int64_t)m * G + d];
        if (p < 0) continue;   // wave-uniform
        unsigned char* dst = payload + (int64_t)(seg_start[d] + p) * row_bytes;
        for (int c = lane * 8; c < K; c += 64 * 8)
            *reinterpret_cast<uint4*>(dst + c * 2) = *reinterpret_cast<const uint4*>(src + c);
        if (lane < topk) {
            const int e = ids[(int64_t)m * topk + lane];
            const int lo = d * epr;
            reinterpret_cast<int*>(dst + (int64_t)K * 2)[lane] = (e >= lo && e < lo + epr) ? e - lo : -1;
            reinterpret_cast<float*>(dst + (int64_t)K * 2 + topk * 4)[lane] = tw[(int64_t)m * topk + lane];
        }
    }
}

// capacity mode: rows [counts[d], capacity) of every segment are not written by ep_pack; give them ids = -1
__global__ __launch_bounds__(256) void ep_pad_kernel(const int* __restrict__ counts, unsigned char* __restrict__ payload,
                                                     int64_t row_bytes, int K, int topk, int G, int capacity) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (row, slot)
    const int64_t row = i / topk;
    const int j = (int)(i - row * topk);
    if (row >= (int64_t)G * capacity) return;
    const int d = (int)(row / capacity);
    if (row - (int64_t)d * capacity >= counts[d]) reinterpret_cast<int*>(payload + row * row_bytes + (int64_t)K * 2)[j] = -1;
}

}  // namespace sglk

using namespace sglk;

extern "C" int sglk_ep_plan(const int32_t* topk_ids, int32_t M, int32_t topk, int32_t E, int32_t G, int32_t capacity,
                            int32_t* counts, int32_t* seg_start, int32_t* pos, int32_t* overflow, void* stream) {
    SGLK_REQUIRE(M >= 0 && topk > 0 && E > 0 && G > 0 && G <= kEpMaxRanks && E % G == 0 && capacity >= 0, SGLK_ERR_INVALID,
                 "ep_plan: bad sizes M=%d topk=%d E=%d G=%d capacity=%d", M, topk, E, G, capacity);
    SGLK_REQUIRE(counts && seg_start && pos && overflow && (M == 0 || topk_ids), SGLK_ERR_INVALID, "ep_plan: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(overflow, 0, sizeof(int), s) != hipSuccess) SGLK_FAIL(SGLK_ERR_LAUNCH, "ep_plan: memset failed");
    const dim3 grid((unsigned)(M > 0 ? ceil_div(M, 1024) : 1));
    const float inv_epr = 1.0f / (float)(E / G);
    if (topk % 4 == 0 && ((uintptr_t)topk_ids % 16) == 0)
        hipLaunchKernelGGL(ep_plan_kernel<true>, grid, dim3(1024), 0, s, topk_ids, M, topk, E, G, inv_epr, capacity, counts, seg_start, pos,
                           overflow);
    else
        hipLaunchKernelGGL(ep_plan_kernel<false>, grid, dim3(1024), 0, s, topk_ids, M, topk, E, G, inv_epr, capacity, counts, seg_start, pos,
                           overflow);
    SGLK_CHECK_LAUNCH("ep_plan");
    return SGLK_OK;
}

extern "C" int sglk_ep_pack(const void* hidden, int64_t hidden_stride, const int32_t* topk_ids, const float* topk_weights,
                            const int32_t* pos, const int32_t* seg_start, const int32_t* counts, void* payload, int64_t row_bytes,
                            int32_t M, int32_t K, int32_t topk, int32_t E, int32_t G, int32_t capacity, void* stream) {
    SGLK_REQUIRE(M >= 0 && K > 0 && topk > 0 && topk <= 64 && G > 0 && G <= kEpMaxRanks && E % G == 0, SGLK_ERR_INVALID,
                 "ep_pack: bad sizes");
    SGLK_REQUIRE(K % 8 == 0 && hidden_stride % 8 == 0 && row_bytes % 16 == 0 && row_bytes >= (int64_t)K * 2 + topk * 8 &&
                     ((uintptr_t)hidden % 16) == 0 && ((uintptr_t)payload % 16) == 0,
                 SGLK_ERR_SHAPE, "ep_pack: rows must be 16-byte aligned (K %% 8 == 0, row_bytes %% 16 == 0)");
    SGLK_REQUIRE(payload && pos && seg_start && counts && (M == 0 || (hidden && topk_ids && topk_weights)), SGLK_ERR_INVALID,
                 "ep_pack: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (M > 0) {
        hipLaunchKernelGGL(ep_pack_kernel, dim3((unsigned)ceil_div(M, 4)), dim3(256), 0, s, (const uint16_t*)hidden, hidden_stride,
                           topk_ids, topk_weights, pos, seg_start, (unsigned char*)payload, row_bytes, M, K, topk, G, E / G);
        SGLK_CHECK_LAUNCH("ep_pack");
    }
    if (capacity > 0) {
        const int64_t items = (int64_t)G * capacity * topk;
        hipLaunchKernelGGL(ep_pad_kernel, dim3((unsigned)ceil_div(items, 256)), dim3(256), 0, s, counts, (unsigned char*)payload,
                           row_bytes, K, topk, G, capacity);
        SGLK_CHECK_LAUNCH("ep_pad");
    }
    return SGLK_OK;
}
