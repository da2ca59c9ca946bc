// One-workgroup stable counting sort of up to 8192 routing slots + the m-tile table (shared by moe_align.hip's one-launch
// variant and by the fused router + align kernel of route_align.hip, whose ids sit in LDS).
#pragma once
#include "sglk_common.h"

namespace sglk {

// Slots are taken in rounds of 1024 (16 waves x 64 lanes, ascending); inside a round a lane's rank among equal experts =
// matches in lower lanes of its wave (ballots) + matches in lower waves (per-wave per-expert counters in LDS).
constexpr int kSmallSlots = 8192;
constexpr int kSmallMaxE = 256;    // 16 waves x 256 experts x 4 B = 16 KiB of wave counters

// `ids` may point to global memory or to LDS (generic pointer); called by ALL 1024 threads of the workgroup
SGLK_DEV void moe_align_small_body(const int* ids, int S, int E, int nbits, int tile_m, int max_tiles,
                                   int* __restrict__ sorted_slot, int* __restrict__ expert_off, int* __restrict__ tile_info,
                                   int* __restrict__ num_tiles) {
    __shared__ int total[kSmallMaxE];          // slots per expert
    __shared__ int off[kSmallMaxE + 1];        // exclusive prefix
    __shared__ int run[kSmallMaxE];            // slots of the expert placed by earlier rounds
    __shared__ int wcnt[16][kSmallMaxE];       // this round: slots of expert e in wave w
    __shared__ int2 scan[2][kSmallMaxE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < kSmallMaxE; e += 1024) { total[e] = 0; run[e] = 0; }
    for (int i = tid; i < 16 * kSmallMaxE; i += 1024) (&wcnt[0][0])[i] = 0;
    __syncthreads();
    for (int s = tid; s < S; s += 1024) {
        const int e = ids[s];
        if (e >= 0 && e < E) atomicAdd(&total[e], 1);
    }
    __syncthreads();
    // scan over experts (E <= 256): threads 0..255
    if (tid < kSmallMaxE) {
        const int t = tid < E ? total[tid] : 0;
        scan[0][tid] = make_int2(t, (t + tile_m - 1) / tile_m);
    }
    __syncthreads();
    int cur = 0;
    for (int d = 1; d < kSmallMaxE; d <<= 1) {
        if (tid < kSmallMaxE) {
            int2 v = scan[cur][tid];
            if (tid >= d) {
                const int2 o = scan[cur][tid - d];
                v.x += o.x;
                v.y += o.y;
            }
            scan[cur ^ 1][tid] = v;
        }
        cur ^= 1;
        __syncthreads();
    }
    if (tid < E) {
        const int2 incl = scan[cur][tid];
        const int t = total[tid];
        const int o = incl.x - t;
        off[tid] = o;
        expert_off[tid] = o;
        if (tid == E - 1) {
            expert_off[E] = incl.x;
            num_tiles[0] = incl.y < max_tiles ? incl.y : max_tiles;
        }
        const int nt = (t + tile_m - 1) / tile_m;
        const int t0 = incl.y - nt;
        for (int i = 0; i < nt && t0 + i < max_tiles; ++i) {
            const int rows = t - i * tile_m < tile_m ? t - i * tile_m : tile_m;
            reinterpret_cast<int4*>(tile_info)[t0 + i] = make_int4(tid, o + i * tile_m, rows, 0);
        }
    }
    __syncthreads();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int base = 0; base < S; base += 1024) {
        const int s = base + tid;
        int e = (s < S) ? ids[s] : -1;
        const bool valid = e >= 0 && e < E;
        if (!valid) e = 0;
        unsigned long long same = __ballot(valid);
        for (int b = 0; b < nbits; ++b) {
            const bool bit = (e >> b) & 1;
            const unsigned long long bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
        const int rank = __popcll(same & lt_mask);
        if (valid && rank == 0) wcnt[wave][e] = __popcll(same);
        __syncthreads();
        if (valid) {
            int before = run[e];
            for (int w = 0; w < wave; ++w) before += wcnt[w][e];
            sorted_slot[off[e] + before + rank] = s;
        }
        __syncthreads();
        // close the round: fold the wave counters into `run` and clear them (one thread per expert)
        if (tid < E) {
            int add = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) { add += wcnt[w][tid]; wcnt[w][tid] = 0; }
            run[tid] += add;
        }
        __syncthreads();
    }
}

}  // namespace sglk
