// moe_align: stable counting sort of the M*topk routing slots by expert + the m-tile table the grouped GEMMs walk.
//
// Replaces the (unobservable) sort/align stage inside the reference's fused_experts_cpu
// (/root/reference/bench_moe.py:113-130); the contract it must honour is the oracle's:
// slot s = m*topk + j belongs to expert topk_ids[s]; ids outside [0,E) are dropped
// (/root/reference/test_moe_offloading_cpu.py:62-68).  Integer-only, bit-reproducible: inside an expert the
// slots keep ascending slot order.
//
// Two launches, no host sync, no atomics on global memory:
//   count : one 256-thread workgroup per 2048-slot chunk, LDS histogram          -> counts[chunk][E]
//   place : one workgroup per chunk.  Every workgroup sums the (small) count table itself -- its chunk's base inside each
//           expert and the experts' totals -- and scans the totals, so no third launch sits between counting and placing;
//           workgroup 0 also writes expert_off[E+1] and the tile table {expert, first position, rows}.  Then each of the four
//           waves places its 512 slots: rank inside the wave by ballot match, running LDS counters across rounds, waves of a
//           chunk offset by the per-wave histograms.
#include <stdlib.h>

#include "fp8_split.h"
#include "quant_rows.h"
#include "knobs.h"
#include "moe_align_small.h"

namespace sglk {

constexpr int kAlignChunk = 2048;   // slots per workgroup (4 waves x 8 rounds x 64 lanes)
constexpr int kMaxExperts = 1024;

__global__ __launch_bounds__(256) void moe_count_kernel(const int* __restrict__ ids, int S, int E,
                                                        int* __restrict__ counts) {
    __shared__ int hist[kMaxExperts];
    const int tid = threadIdx.x;
    for (int e = tid; e < E; e += 256) hist[e] = 0;
    __syncthreads();
    const int base = blockIdx.x * kAlignChunk;
#pragma unroll
    for (int r = 0; r < kAlignChunk / 256; ++r) {
        const int s = base + r * 256 + tid;
        if (s < S) {
            const int e = ids[s];
            if (e >= 0 && e < E) atomicAdd(&hist[e], 1);
        }
    }
    __syncthreads();
    for (int e = tid; e < E; e += 256) counts[(size_t)blockIdx.x * E + e] = hist[e];
}

// tail_max > 0: the last of an expert's SEVERAL tiles goes to a second table (tile_info_b / num_tiles_b) when it has at most tail_max rows
// -- fused_experts runs those tail tiles on the weight-streaming mid kernel instead of paying a full 256-row tile for them.
// zero16: sixteen ints the caller wants cleared before its next launch (the tile tickets of the persistent GEMMs), or null.
__global__ __launch_bounds__(256) void moe_place_kernel(const int* __restrict__ ids, int S, int E, int nbits,
                                                        const int* __restrict__ counts, int nchunk, int tile_m, int max_tiles,
                                                        int* __restrict__ sorted_slot, int* __restrict__ expert_off,
                                                        int* __restrict__ tile_info, int* __restrict__ num_tiles, int tail_max,
                                                        int* __restrict__ tile_info_b, int* __restrict__ num_tiles_b,
                                                        int* __restrict__ zero16, const SplitJob job, int place_blocks) {
    // workgroups past the placing ones split rows of `hidden` for the two-term W8A16 kernels (fp8_split.h): independent work
    // that would otherwise be a launch of its own behind this one
    if ((int)blockIdx.x >= place_blocks) {
        const int64_t row = (int64_t)((int)blockIdx.x - place_blocks) * 4 + (threadIdx.x >> 6);
        if (row < job.rows) {
            if (job.terms == 0)
                quant_row_int8(job.x + row * job.x_stride, (int8_t*)(job.q + row * job.q_stride), job.sf + row, job.cols, job.floor_v, threadIdx.x & 63);
            else if (job.terms == 2)
                split_row_block128(job.x + row * job.x_stride, job.q + row * job.q_stride, job.s + row * job.s_stride, job.cols, threadIdx.x & 63);
            else
                quant_row_block128(job.x + row * job.x_stride, job.q + row * job.q_stride, job.s + row * job.s_stride, job.cols, threadIdx.x & 63);
        }
        return;
    }
    __shared__ int s_tot[kMaxExperts];       // slots of the expert in the whole input
    __shared__ int s_base[kMaxExperts];      // slots of the expert in earlier chunks; after the scan: first position of MY chunk's
    __shared__ int s_run[4][kMaxExperts];    // per wave: slots of the expert placed so far by this chunk
    __shared__ int4 s_scan[2][256];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int blk = blockIdx.x;
    for (int e = tid; e < E; e += 256) {
        s_tot[e] = 0;
        s_base[e] = 0;
        s_run[0][e] = s_run[1][e] = s_run[2][e] = s_run[3][e] = 0;
    }
    if (blk == 0 && zero16 && tid < 16) zero16[tid] = 0;
    __syncthreads();
    // my slots' ids (kept in registers) and the per-wave histograms of this chunk
    const int first = blk * kAlignChunk + wv * (kAlignChunk / 4);
    int my[kAlignChunk / 256];
#pragma unroll
    for (int r = 0; r < kAlignChunk / 256; ++r) {
        const int s = first + r * 64 + lane;
        int e = (s < S) ? ids[s] : -1;
        if (e < 0 || e >= E) e = -1;
        my[r] = e;
        if (e >= 0) atomicAdd(&s_run[wv][e], 1);
    }
    // per expert: sum of the count table over all chunks and over the chunks in front of mine (`parts` threads share a column)
    {
        const int parts = E < 256 ? 256 / E : 1;
        const int per = (nchunk + parts - 1) / parts;
        for (int idx = tid; idx < E * parts; idx += 256) {
            const int e = idx % E, part = idx / E;
            const int c0 = part * per, c1 = (c0 + per < nchunk) ? c0 + per : nchunk;
            int tot = 0, before = 0;
#pragma unroll 8
            for (int c = c0; c < c1; ++c) {
                const int v = counts[(size_t)c * E + e];
                tot += v;
                before += c < blk ? v : 0;
            }
            atomicAdd(&s_tot[e], tot);
            atomicAdd(&s_base[e], before);
        }
    }
    __syncthreads();
    // exclusive scan over experts of {slots, tiles, tail tiles}: a thread owns `own` consecutive experts
    const int own = (E + 255) / 256;
    const int e0 = tid * own;
    int4 mine = make_int4(0, 0, 0, 0);
    for (int i = 0; i < own; ++i) {
        const int e = e0 + i;
        if (e < E) {
            const int total = s_tot[e];
            const int nt_all = (total + tile_m - 1) / tile_m;
            const int rem = total - (nt_all - 1) * tile_m;         // rows of the expert's last tile (nt_all > 0)
            const int tail = (tail_max > 0 && nt_all >= 2 && rem <= tail_max) ? 1 : 0;   // a true tail: the expert has full tiles too
            mine.x += total;
            mine.y += nt_all - tail;
            mine.z += tail;
        }
    }
    s_scan[0][tid] = mine;
    __syncthreads();
    int cur = 0;
    for (int d = 1; d < 256; d <<= 1) {
        int4 v = s_scan[cur][tid];
        if (tid >= d) {
            const int4 o = s_scan[cur][tid - d];
            v.x += o.x;
            v.y += o.y;
            v.z += o.z;
        }
        s_scan[cur ^ 1][tid] = v;
        cur ^= 1;
        __syncthreads();
    }
    const int4 incl = s_scan[cur][tid];
    int4 run = make_int4(incl.x - mine.x, incl.y - mine.y, incl.z - mine.z, 0);
    for (int i = 0; i < own; ++i) {
        const int e = e0 + i;
        if (e >= E) break;
        const int total = s_tot[e];
        const int nt_all = (total + tile_m - 1) / tile_m;
        const int rem = total - (nt_all - 1) * tile_m;
        const int tail = (tail_max > 0 && nt_all >= 2 && rem <= tail_max) ? 1 : 0;
        const int off = run.x, nt = nt_all - tail;
        s_base[e] += off;
        if (blk == 0) {
            expert_off[e] = off;
            for (int t = 0; t < nt && run.y + t < max_tiles; ++t) {
                const int rows = total - t * tile_m < tile_m ? total - t * tile_m : tile_m;
                reinterpret_cast<int4*>(tile_info)[run.y + t] = make_int4(e, off + t * tile_m, rows, 0);
            }
            if (tail) reinterpret_cast<int4*>(tile_info_b)[run.z] = make_int4(e, off + (nt_all - 1) * tile_m, rem, 0);
        }
        run.x += total;
        run.y += nt;
        run.z += tail;
    }
    if (blk == 0 && tid == 255) {
        expert_off[E] = incl.x;
        num_tiles[0] = incl.y < max_tiles ? incl.y : max_tiles;
        if (num_tiles_b) num_tiles_b[0] = incl.z;
    }
    // per-wave histograms -> slots of the expert in LOWER waves of this chunk
    for (int e = tid; e < E; e += 256) {
        const int a0 = s_run[0][e], a1 = s_run[1][e], a2 = s_run[2][e];
        s_run[0][e] = 0;
        s_run[1][e] = a0;
        s_run[2][e] = a0 + a1;
        s_run[3][e] = a0 + a1 + a2;
    }
    __syncthreads();
    // place: a wave walks its 512 slots in 8 rounds; a wave's LDS instructions execute in order for all its lanes, so the
    // read of run[e] by every lane precedes the leader's update without a barrier
    volatile int* runw = s_run[wv];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < kAlignChunk / 256; ++r) {
        const int s = first + r * 64 + lane;
        const bool valid = my[r] >= 0;
        const int e = valid ? my[r] : 0;
        // lanes holding the same expert id (emulated match-any: one ballot per id bit)
        unsigned long long same = __ballot(valid);
        for (int b = 0; b < nbits; ++b) {
            const bool bit = (e >> b) & 1;
            const unsigned long long bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
        const int rank = __popcll(same & lt_mask);
        const int before = valid ? runw[e] : 0;
        if (valid && rank == 0) runw[e] = before + __popcll(same);
        if (valid) sorted_slot[s_base[e] + before + rank] = s;
    }
}

// ---- one-launch variant for small inputs (S <= kSmallSlots): a single 1024-thread workgroup (moe_align_small.h) ----------
// Same outputs, same stable order.  Launch-bound sizes: replaces three dependent launches by one.
__global__ __launch_bounds__(1024) void moe_align_small_kernel(const int* __restrict__ ids, int S, int E, int nbits,
                                                               int tile_m, int max_tiles, int* __restrict__ sorted_slot,
                                                               int* __restrict__ expert_off, int* __restrict__ tile_info,
                                                               int* __restrict__ num_tiles, int* __restrict__ zero16) {
    if (zero16 && threadIdx.x < 16) zero16[threadIdx.x] = 0;
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
                           void* workspace, size_t workspace_bytes, void* stream, int32_t* zero16, const SplitJob* job,
                           bool* job_taken) {
    if (job_taken) *job_taken = false;
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
    // one launch for small inputs -- unless a row job wants to ride in the placing launch (then count + place are two launches
    // either way and the job's own launch is saved), and only up to 6144 slots: at 8192 the single 1024-thread workgroup takes
    // 28 us against 20 us for count + place (profiles/r03_v3_stage_sweep.json: M = 1024 vs 1536)
    const bool has_job = job && job->rows > 0;
    if (tail_max <= 0 && !has_job && S <= 6144 && S <= kSmallSlots && E <= kSmallMaxE && !knobs().align_3pass) {
        hipLaunchKernelGGL(moe_align_small_kernel, dim3(1), dim3(1024), 0, s, topk_ids, S, E, nbits, tile_m, max_tiles,
                           sorted_slot, expert_off, tile_info, num_tiles, zero16);
        SGLK_CHECK_LAUNCH("moe_align");
        return SGLK_OK;
    }
    if (nchunk > 0) hipLaunchKernelGGL(moe_count_kernel, dim3(nchunk), dim3(256), 0, s, topk_ids, S, E, counts);
    const int place_blocks = nchunk > 0 ? nchunk : 1;
    SplitJob j{};
    int split_blocks = 0;
    if (job && job->rows > 0) {
        j = *job;
        split_blocks = (int)ceil_div(job->rows, 4);
        if (job_taken) *job_taken = true;
    }
    hipLaunchKernelGGL(moe_place_kernel, dim3(place_blocks + split_blocks), dim3(256), 0, s, topk_ids, S, E, nbits, counts, nchunk, tile_m,
                       max_tiles, sorted_slot, expert_off, tile_info, num_tiles, tail_max > 0 ? tail_max : 0, tile_info_b, num_tiles_b,
                       zero16, j, place_blocks);
    SGLK_CHECK_LAUNCH("moe_align");
    return SGLK_OK;
}
}  // namespace sglk

extern "C" int sglk_moe_align(const int32_t* topk_ids, int32_t M, int32_t E, int32_t topk, int32_t tile_m,
                              int32_t* sorted_slot, int32_t* expert_off, int32_t* tile_info, int32_t* num_tiles,
                              void* workspace, size_t workspace_bytes, void* stream) {
    return sglk::launch_moe_align_split(topk_ids, M, E, topk, tile_m, sorted_slot, expert_off, tile_info, num_tiles, 0, nullptr,
                                        nullptr, workspace, workspace_bytes, stream, nullptr, nullptr, nullptr);
}
