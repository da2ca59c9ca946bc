// moe_align for a handful of slots, inside the consuming kernel: at decode sizes (M * topk <= 32) the align launch costs as much as
// the GEMM it feeds, and its whole result -- which expert a workgroup streams, which slots are that expert's rows -- is a
// function of <= 32 integers that every wave can compute for itself.
//   tiles    = the distinct valid experts in ascending id order (one tile each: an expert has at most 32 rows here)
//   position = rank of a slot in (expert id, slot index) order over the valid slots: ic1 row of the slot, the same in GEMM-1
//              and GEMM-2 because both run these functions on the same ids
// ids outside [0, E) are skipped like moe_align skips them (/root/reference/test_moe_offloading_cpu.py:62-68).
#pragma once
#include "sglk_common.h"

namespace sglk {

constexpr int kInlineAlignSlots = 32;
constexpr int kInlineNoExpert = 0x7fffffff;

struct InlineAlign {
    int id;                       // this lane's expert (kInlineNoExpert: no slot / masked id)
    int tile;                     // tile index of this lane's expert = distinct valid experts below it
    unsigned long long firsts;    // lanes that are the lowest slot of their expert
    int ntiles;
};

// one wave, lane = slot index, S <= 32.  Straight-line: the ids of the other slots arrive by v_readlane with a literal lane, one
// ballot tells whether slot j is the first of its expert, and only those count towards the tile index (32 x ~8 instructions; a
// loop with a scalar lane index and branches took several microseconds and ate the launch it replaced).
SGLK_DEV InlineAlign inline_align(const int* __restrict__ ids, int S, int E, int lane) {
    InlineAlign r;
    const int raw = lane < S ? ids[lane] : -1;
    r.id = (unsigned)raw < (unsigned)E ? raw : kInlineNoExpert;
    r.tile = 0;
    r.firsts = 0;
#pragma unroll
    for (int j = 0; j < kInlineAlignSlots; ++j) {
        if (j >= S) break;
        const int idj = __builtin_amdgcn_readlane(r.id, j);
        const unsigned long long same = __ballot(r.id == idj);
        const bool first_j = idj != kInlineNoExpert && (same & ((1ull << j) - 1ull)) == 0ull;   // wave-uniform
        if (first_j) {
            r.firsts |= 1ull << j;
            r.tile += idj < r.id ? 1 : 0;
        }
    }
    r.ntiles = __popcll(r.firsts);
    return r;
}

struct InlineTile { int e, pos0, rows; };

// the tile `mtile` (< ntiles); lanes of that expert write their slot index at their row of slot_tab (wave 0 only)
SGLK_DEV InlineTile inline_tile(const InlineAlign& a, int mtile, int lane, bool writer, int* slot_tab) {
    InlineTile t;
    const unsigned long long hit = __ballot(((a.firsts >> lane) & 1ull) && a.tile == mtile);
    const int src = __ffsll(hit) - 1;
    t.e = __builtin_amdgcn_readlane(a.id, src);
    t.pos0 = __popcll(__ballot(a.id < t.e));
    const unsigned long long same = __ballot(a.id == t.e);
    t.rows = __popcll(same);
    if (writer && a.id == t.e) slot_tab[__popcll(same & ((1ull << lane) - 1ull))] = lane;
    return t;
}

}  // namespace sglk
