// moe_align: stable counting sort of the M*topk routing slots by expert + the m-tile table the grouped GEMMs walk.
//
// Replaces the (unobservable) sort/align stage inside the reference's fused_experts_cpu
// (/root/reference/bench_moe.py:113-130); the contract it must honour is the oracle's:
// slot s = m*topk + j belongs to expert topk_ids[s]; ids outside [0,E) are dropped
// (/root/reference/test_moe_offloading_cpu.py:62-68).  Integer-only, bit-reproducible: inside an expert the
// slots keep ascending slot order.
//
// Three launches, no host sync, no atomics on global memory:
//   count   : one wave per 512-slot chunk, LDS histogram                  -> counts[chunk][E]
//   scan    : one workgroup: per-expert running sum over chunks (in place -> chunk base), block scan over
//             experts -> expert_off[E+1]; tile table {expert, first position, rows} for tile_m rows per tile
//   scatter : one wave per chunk, rank inside the wave by ballot match, running LDS counters across rounds
#include <stdlib.h>

#include "knobs.h"
#include "moe_align_small.h"

namespace sglk {

constexpr int kAlignChunk = 512;   // slots per (single-wave) workgroup
constexpr int kMaxExperts = 1024;

__global__ __launch_bounds__(64) void moe_count_kernel(const int* __restrict__ ids, int S, int E,
                                                       int* __restrict__ counts) {
    __shared__ int hist[kMaxExperts];
    const int lane = threadIdx.x;
    for (int e = lane; e < E; e += 64) hist[e] = 0;
    __syncthreads();
    const int base = blockIdx.x * kAlignChunk;
#pragma unroll
    for (int r = 0; r < kAlignChunk / 64; ++r) {
        const int s = base + r * 64 + lane;
        if (s < S) {
            const int e = ids[s];
            if (e >= 0 && e < E) atomicAdd(&hist[e], 1);
        }
    }
    __syncthreads();
    for (int e = lane; e < E; e += 64) counts[(size_t)blockIdx.x * E + e] = hist[e];
}

// tail_max > 0: the last of an expert's SEVERAL tiles goes to a second table (tile_info_b / num_tiles_b) when it has at most tail_max rows
// -- fused_experts runs those tail tiles on the weight-streaming mid kernel instead of paying a full 256-row tile for them.
__global__ __launch_bounds__(1024) void moe_scan_kernel(int* __restrict__ counts, int nchunk, int E, int tile_m,
                                                        int max_tiles, int* __restrict__ expert_off,
                                                        int* __restrict__ tile_info, int* __restrict__ num_tiles,
                                                        int tail_max, int* __restrict__ tile_info_b,
                                                        int* __restrict__ num_tiles_b) {
    __shared__ int part_sum[kMaxExperts];   // [part][expert], parts * E <= 1024
    __shared__ int cnt[kMaxExperts];
    __shared__ int4 scan[2][kMaxExperts];   // {slots, tiles, tail tiles} inclusive scans, ping-pong
    const int tid = threadIdx.x;
    // `parts` threads share one expert's column of the [chunk][expert] table, each walking a contiguous chunk range
    const int parts = kMaxExperts / E;
    const int e_of = tid % E, part = tid / E;
    const int per = (nchunk + parts - 1) / parts;
    const int c0 = part * per, c1 = (c0 + per < nchunk) ? c0 + per : nchunk;
    int local = 0;
    if (part < parts) {
#pragma unroll 4
        for (int c = c0; c < c1; ++c) local += counts[(size_t)c * E + e_of];
        part_sum[part * E + e_of] = local;
    }
    __syncthreads();
    int total = 0;
    if (part < parts) {
        int base = 0;
        for (int q = 0; q < parts; ++q) {
            const int v = part_sum[q * E + e_of];
            if (q < part) base += v;
            total += v;
        }
        // second walk: counts[c][e] becomes the chunk's base inside the expert (exclusive running sum)
        int run = base;
#pragma unroll 4
        for (int c = c0; c < c1; ++c) {
            const int t = counts[(size_t)c * E + e_of];
            counts[(size_t)c * E + e_of] = run;
            run += t;
        }
    }
    const int e = tid;
    if (part == 0) cnt[e_of] = total;
    __syncthreads();
    total = (e < E) ? cnt[e] : 0;
    const int nt_all = (total + tile_m - 1) / tile_m;
    const int rem = total - (nt_all - 1) * tile_m;             // rows of the expert's last tile (nt_all > 0)
    const int tail = (tail_max > 0 && nt_all >= 2 && rem <= tail_max) ? 1 : 0;   // a true tail: the expert has full tiles too
    scan[0][e] = make_int4(total, nt_all - tail, tail, 0);
    __syncthreads();
    int cur = 0;
    for (int d = 1; d < kMaxExperts; d <<= 1) {
        int4 v = scan[cur][e];
        if (e >= d) {
            const int4 o = scan[cur][e - d];
            v.x += o.x;
            v.y += o.y;
            v.z += o.z;
        }
        scan[cur ^ 1][e] = v;
        cur ^= 1;
        __syncthreads();
    }
    const int4 incl = scan[cur][e];
    if (e < E) {
        const int off = incl.x - total;
        expert_off[e] = off;
        if (e == E - 1) {
            expert_off[E] = incl.x;
            num_tiles[0] = incl.y < max_tiles ? incl.y : max_tiles;
            if (num_tiles_b) num_tiles_b[0] = incl.z;
        }
        const int nt = nt_all - tail;
        int t0 = incl.y - nt;
        for (int i = 0; i < nt && t0 + i < max_tiles; ++i) {
            const int rows = total - i * tile_m < tile_m ? total - i * tile_m : tile_m;
            reinterpret_cast<int4*>(tile_info)[t0 + i] = make_int4(e, off + i * tile_m, rows, 0);
        }
        if (tail) reinterpret_cast<int4*>(tile_info_b)[incl.z - 1] = make_int4(e, off + (nt_all - 1) * tile_m, rem, 0);
    }
}

__global__ __launch_bounds__(64) void moe_scatter_kernel(const int* __restrict__ ids, int S, int E, int nbits,
                                                         const int* __restrict__ chunk_base,
                                                         const int* __restrict__ expert_off,
                                                         int* __restrict__ sorted_slot) {
    __shared__ int run[kMaxExperts];
    const int lane = threadIdx.x;
    for (int e = lane; e < E; e += 64) run[e] = 0;
    __syncthreads();
    const int base = blockIdx.x * kAlignChunk;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int r = 0; r < kAlignChunk / 64; ++r) {
        const int s = base + r * 64 + lane;
        int e = (s < S) ? ids[s] : -1;
        const bool valid = e >= 0 && e < E;
        if (!valid) e = 0;
        // lanes holding the same expert id (emulated match-any: one ballot per id bit)
        unsigned long long same = __ballot(valid);
        for (int b = 0; b < nbits; ++b) {
            const bool bit = (e >> b) & 1;
            const unsigned long long bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
        const int rank = __popcll(same & lt_mask);
        const int before = valid ? run[e] : 0;
        __syncthreads();
        if (valid && rank == 0) run[e] = before + __popcll(same);
        __syncthreads();
        if (valid) {
            const int pos = expert_off[e] + chunk_base[(size_t)blockIdx.x * E + e] + before + rank;
            sorted_slot[pos] = s;
        }
    }
}

// ---- one-launch variant for small inputs (S <= kSmallSlots): a single 1024-thread workgroup (moe_align_small.h) ----------
// Same outputs, same stable order.  Launch-bound sizes: replaces three dependent launches by one.
__global__ __launch_bounds__(1024) void moe_align_small_kernel(const int* __restrict__ ids, int S, int E, int nbits,
                                                               int tile_m, int max_tiles, int* __restrict__ sorted_slot,
                                                               int* __restrict__ expert_off, int* __restrict__ tile_info,
                                                               int* __restrict__ num_tiles) {
    moe_align_small_body(ids, S, E, nbits, tile_m, max_tiles, sorted_slot, expert_off, tile_info, num_tiles);
}

}  // namespace sglk

using namespace sglk;

extern "C" size_t sglk_moe_align_workspace_bytes(int32_t M, int32_t E, int32_t topk) {
    if (M < 0 || E <= 0 || topk <= 0) return 0;
    const int64_t S = (int64_t)M * topk;
    const int64_t nchunk = ceil_div(S, kAlignChunk);
    return align_up((size_t)(nchunk > 0 ? nchunk : 1) * E * sizeof(int), 256);
}

extern "C" int32_t sglk_moe_max_tiles(int32_t M, int32_t E, int32_t topk, int32_t tile_m) {
    if (M <= 0 || E <= 0 || topk <= 0 || tile_m <= 0) return 0;
    const int64_t S = (int64_t)M * topk;
    // every expert can add at most one partial tile, and no more experts than slots can be hit
    const int64_t partial = S < E ? S : E;
    return (int32_t)(S / tile_m + partial);
}

namespace sglk {
int launch_moe_align_split(const int32_t* topk_ids, int32_t M, int32_t E, int32_t topk, int32_t tile_m,
                           int32_t* sorted_slot, int32_t* expert_off, int32_t* tile_info, int32_t* num_tiles,
                           int32_t tail_max, int32_t* tile_info_b, int32_t* num_tiles_b,
                           void* workspace, size_t workspace_bytes, void* stream) {
    SGLK_REQUIRE(M >= 0 && E > 0 && topk > 0 && tile_m > 0, SGLK_ERR_INVALID, "moe_align: bad sizes M=%d E=%d topk=%d", M, E, topk);
    SGLK_REQUIRE(E <= kMaxExperts, SGLK_ERR_SHAPE, "moe_align: at most %d experts supported (got %d)", kMaxExperts, E);
    SGLK_REQUIRE((int64_t)M * topk < (1ll << 31), SGLK_ERR_SHAPE, "moe_align: M*topk overflows int32");
    SGLK_REQUIRE(expert_off && tile_info && num_tiles && workspace, SGLK_ERR_INVALID, "moe_align: null pointer");
    SGLK_REQUIRE(M == 0 || (topk_ids && sorted_slot), SGLK_ERR_INVALID, "moe_align: null pointer");
    SGLK_REQUIRE(workspace_bytes >= sglk_moe_align_workspace_bytes(M, E, topk), SGLK_ERR_WORKSPACE,
                 "moe_align: workspace too small");
    SGLK_REQUIRE(tail_max <= 0 || (tile_info_b && num_tiles_b), SGLK_ERR_INVALID, "moe_align: tail table missing");
    hipStream_t s = (hipStream_t)stream;
    const int S = M * topk;
    const int nchunk = (int)ceil_div(S, kAlignChunk);
    int* counts = (int*)workspace;
    int nbits = 0;
    while ((1 << nbits) < E) ++nbits;
    const int max_tiles = sglk_moe_max_tiles(M, E, topk, tile_m);
    if (tail_max <= 0 && S <= kSmallSlots && E <= kSmallMaxE && !knobs().align_3pass) {
        hipLaunchKernelGGL(moe_align_small_kernel, dim3(1), dim3(1024), 0, s, topk_ids, S, E, nbits, tile_m, max_tiles,
                           sorted_slot, expert_off, tile_info, num_tiles);
        SGLK_CHECK_LAUNCH("moe_align");
        return SGLK_OK;
    }
    if (nchunk > 0) {
        hipLaunchKernelGGL(moe_count_kernel, dim3(nchunk), dim3(64), 0, s, topk_ids, S, E, counts);
    }
    hipLaunchKernelGGL(moe_scan_kernel, dim3(1), dim3(kMaxExperts), 0, s, counts, nchunk, E, tile_m, max_tiles,
                       expert_off, tile_info, num_tiles, tail_max > 0 ? tail_max : 0, tile_info_b, num_tiles_b);
    if (nchunk > 0) {
        hipLaunchKernelGGL(moe_scatter_kernel, dim3(nchunk), dim3(64), 0, s, topk_ids, S, E, nbits, counts,
                           expert_off, sorted_slot);
    }
    SGLK_CHECK_LAUNCH("moe_align");
    return SGLK_OK;
}
}  // namespace sglk

extern "C" int sglk_moe_align(const int32_t* topk_ids, int32_t M, int32_t E, int32_t topk, int32_t tile_m,
                              int32_t* sorted_slot, int32_t* expert_off, int32_t* tile_info, int32_t* num_tiles,
                              void* workspace, size_t workspace_bytes, void* stream) {
    return sglk::launch_moe_align_split(topk_ids, M, E, topk, tile_m, sorted_slot, expert_off, tile_info, num_tiles, 0, nullptr,
                                        nullptr, workspace, workspace_bytes, stream);
}
