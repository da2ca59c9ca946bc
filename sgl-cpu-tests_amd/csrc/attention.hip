// extend_attention / decode_attention: paged-KV flash attention on bf16 MFMA.
//
// Replaces torch.ops.sgl_kernel.extend_attention_cpu (/root/reference/test_extend.py:168-182, bench_extend.py:70-102)
// and decode_attention_cpu (/root/reference/test_mla.py:115-128, test_decoding.py:107-120).  Oracles:
// _run_sdpa_forward_extend (/root/reference/test_extend.py:10-76) and _run_sdpa_forward_decode (test_mla.py:12-66).
//
// One device core serves both.  A wave owns QT tiles of 16 "query columns" that share one K/V stream:
//   extend : columns = consecutive query positions of one (sequence, q head);  keys = paged prefix (all visible) then
//            the extend tokens (causal)
//   decode : columns = the q heads of one kv head (GQA / MLA group), one position;  keys = one split of the sequence
// Per 64-key tile (K and V staged in LDS once per workgroup):
//   S^T = K . Q^T   keys are the MFMA rows, queries the columns -> a lane owns ONE query column: its running max / sum
//                    are plain per-lane scalars and the only cross-lane traffic is two shuffles per tile for the max;
//   P = exp(S - m)  stays in the accumulator registers and IS the B operand of the next product (16 keys of tile 2s
//                    and 16 of tile 2s+1 form the 32-deep k-step; the matching A operand rows of V come from
//                    ds_read_b64_tr_b16, which transposes 4 keys x 16 columns out of the row-major V image);
//   O^T += V^T . P
// K rows are 16-byte-chunk XOR-swizzled in LDS (conflict-free ds_read_b128); when V aliases K (MLA: v_buffer is
// k_buffer[..., :DV], /root/reference/test_mla.py:83) the K image is reused for V and K is read from HBM once.
// fp32 softmax, bf16 P, fp32 accumulation; optional logit soft-cap.  Decode writes per-split (O/l, lse) into the
// caller's attn_logits [B][HQ][splits][DV+1] and a merge kernel combines them.
#include "knobs.h"

namespace sglk {
namespace attn {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

constexpr int kKeys = 64;   // keys per tile

template <int CHUNKS>
struct Swz {   // XOR mask for 16-byte chunks of a row with CHUNKS chunks (mask+1 must divide CHUNKS)
    // a ds_read_b128 is serviced in groups of SIXTEEN lanes ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md 'LDS'), i.e. 16 rows of
    // two neighbouring chunk columns in the K-fragment read: only a 4-bit mask spreads those over all 64 banks (with 3 bits the
    // two columns share one 128-byte window: a 2-way conflict on every fragment read).  Rows whose chunk count is not a multiple
    // of 16 (D = 192, 576) keep the 3-bit mask.
    static constexpr int mask = (CHUNKS % 16 == 0) ? 15 : ((CHUNKS % 8 == 0) ? 7 : ((CHUNKS % 4 == 0) ? 3 : ((CHUNKS % 2 == 0) ? 1 : 0)));
};

struct KvSource {
    // key position p of the stream -> row pointer.  p < n_paged: buf + page[p] * stride_tok;  else ext + (p - n_paged) * ext_stride_tok
    const unsigned short* buf;
    int64_t buf_stride_tok;
    const void* page;     // int32 or int64 token ids
    int page_is64;
    int n_paged;
    const unsigned short* ext;
    int64_t ext_stride_tok;
};

SGLK_DEV const unsigned short* kv_row(const KvSource& s, int p) {
    if (p < s.n_paged) {
        const int64_t tok = s.page_is64 ? reinterpret_cast<const int64_t*>(s.page)[p] : (int64_t)reinterpret_cast<const int*>(s.page)[p];
        return s.buf + tok * s.buf_stride_tok;
    }
    return s.ext + (int64_t)(p - s.n_paged) * s.ext_stride_tok;
}
// Every position is paged (decode).  Branch-free on purpose: a branch around the page lookup makes the compiler wait
// for every lookup before the next one is issued, which serialises a tile's 4..18 row fetches per lane (measured:
// 13.6 us per 64-key MLA tile instead of ~3).
SGLK_DEV const unsigned short* kv_row_paged(const KvSource& s, int p) {
    const int64_t tok = s.page_is64 ? reinterpret_cast<const int64_t*>(s.page)[p] : (int64_t)reinterpret_cast<const int*>(s.page)[p];
    return s.buf + tok * s.buf_stride_tok;
}

// stage `nkeys` (<= 64) rows of WIDTH bf16 into the swizzled LDS image; missing rows are zero-filled
template <int WIDTH, int THREADS>
SGLK_DEV void stage_tile(unsigned char* lds, const KvSource& src, int p0, int nkeys, const unsigned short** rowptr_lds) {
    constexpr int CH = WIDTH / 8;
    constexpr int MASK = Swz<CH>::mask;
    if (threadIdx.x < kKeys) rowptr_lds[threadIdx.x] = threadIdx.x < nkeys ? kv_row(src, p0 + threadIdx.x) : nullptr;
    __syncthreads();
    for (int c = threadIdx.x; c < kKeys * CH; c += THREADS) {
        const int row = c / CH, ch = c - row * CH;
        const unsigned short* rp = rowptr_lds[row];
        uint4 v = make_uint4(0, 0, 0, 0);
        if (rp) v = *reinterpret_cast<const uint4*>(rp + ch * 8);
        *reinterpret_cast<uint4*>(lds + row * (WIDTH * 2) + ((ch ^ (row & MASK)) << 4)) = v;
    }
}

// Split staging for software pipelining (issue the global loads of tile i+1 before computing tile i, write them to the
// other LDS buffer afterwards).  Every thread resolves the rows of its own chunks (page lookups hit L1), so no
// row-pointer table and no extra barrier is needed.  NCH = chunks per thread.
template <int WIDTH, int THREADS>
struct TileRegs {
    static constexpr int CH = WIDTH / 8;
    static constexpr int NCH = (kKeys * CH + THREADS - 1) / THREADS;
    uint4 v[NCH];
    // real_w: valid elements per row (<= WIDTH, even; the image is zero-padded beyond it); aligned: rows and their
    // 8-element chunks are 16-byte aligned (else the chunk is gathered as four dwords)
    SGLK_DEV void load(const KvSource& src, int p0, int nkeys, int real_w = WIDTH, bool aligned = true) {
        // two passes: row pointers (page lookups) first, then the row loads back to back
        const unsigned short* rp[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = threadIdx.x + i * THREADS;
            const int row = c / CH, ch = c - row * CH;
            rp[i] = (c < kKeys * CH && row < nkeys && ch * 8 < real_w) ? kv_row(src, p0 + row) + ch * 8 : nullptr;
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            v[i] = make_uint4(0, 0, 0, 0);
            if (!rp[i]) continue;
            const int ch = (threadIdx.x + i * THREADS) % CH;
            if (aligned && ch * 8 + 8 <= real_w) {
                v[i] = *reinterpret_cast<const uint4*>(rp[i]);
            } else {
                const unsigned* d = reinterpret_cast<const unsigned*>(rp[i]);
                const int left = real_w - ch * 8;            // 2, 4, 6 or >= 8 valid elements
                v[i].x = d[0];
                if (left > 2) v[i].y = d[1];
                if (left > 4) v[i].z = d[2];
                if (left > 6) v[i].w = d[3];
            }
        }
    }
    SGLK_DEV void store(unsigned char* lds) const {
        constexpr int MASK = Swz<CH>::mask;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = threadIdx.x + i * THREADS;
            const int row = c / CH, ch = c - row * CH;
            if (c < kKeys * CH) *reinterpret_cast<uint4*>(lds + row * (WIDTH * 2) + ((ch ^ (row & MASK)) << 4)) = v[i];
        }
    }
};

// Per-wave attention state and the per-tile update.  D, DV: head dims of K and V; QT: 16-column query tiles.
template <int D, int DV, int QT>
struct Core {
    static constexpr int KS = D / 32;       // k-steps of Q.K
    static constexpr int VT = DV / 16;      // output tiles
    static constexpr int KCH = D / 8, VCH = DV / 8;

    bf16x8 qf[QT][KS];
    f32x4 o[QT][VT];
    float m[QT], l[QT];

    SGLK_DEV void init() {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            m[qt] = -INFINITY;
            l[qt] = 0.f;
#pragma unroll
            for (int t = 0; t < VT; ++t) o[qt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    // column qt*16 + (lane&15) reads its query row (nullptr = padding column)
    // d_real: valid elements of the row (multiple of 8, <= D; zero beyond)
    SGLK_DEV void load_q(int qt, const unsigned short* qrow, int lane, int d_real = D) {
        const int g = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (qrow && ks * 32 + g * 8 < d_real) v = *reinterpret_cast<const uint4*>(qrow + ks * 32 + g * 8);
            qf[qt][ks] = __builtin_bit_cast(bf16x8, v);
        }
    }

    // one 64-key tile.  key_base: stream position of the tile's first key; limit[qt]: the lane's column may see keys
    // with position < limit; V_ALIAS: V rows live in the K image (row width D), else in its own image (row width DV)
    // DV here is the width of the V slice this wave accumulates; dv0 = its first column inside the V image, whose rows
    // are VROW elements wide (0 = DV, i.e. the wave owns the whole width)
    // NKT = 16-key tiles this wave takes from the image (4 = all 64 keys; 2 = the 32 keys from row k0, decode kernel)
    // ASM_TR: the transposed V reads are inline asm with hand-counted lgkmcnt waits.  The compiler does not know what the
    // ds_read_tr builtin may alias and puts an s_waitcnt vmcnt(0) in front of the first one whenever LDS-DMA writes are pending:
    // in the decode kernel that drained the NEXT tiles' rows in the middle of every tile.
    template <bool V_ALIAS, int VROW = 0, int NKT = 4, bool ASM_TR = false>
    SGLK_DEV void tile(const unsigned char* klds, const unsigned char* vlds, int key_base, const int (&limit)[QT],
                       float scale_log2e, float logit_cap, int lane, int dv0 = 0, int k0 = 0) {
        int lim_min = limit[0];
#pragma unroll
        for (int qt = 1; qt < QT; ++qt) lim_min = limit[qt] < lim_min ? limit[qt] : lim_min;
        const bool masked = __any(key_base + k0 + NKT * 16 > lim_min);
        const bool capped = __builtin_amdgcn_readfirstlane(logit_cap > 0.f);
        constexpr int KMASK = Swz<KCH>::mask;
        constexpr int VW = V_ALIAS ? D : (VROW ? VROW : DV);   // row width of the image V is read from
        constexpr int VMASK = V_ALIAS ? KMASK : Swz<VW / 8>::mask;
        constexpr float kRescaleThr = 8.0f;                  // log2 units: P <= 2^8 before a rescale is forced
        const int r = lane & 15, g = lane >> 4;
        f32x4 s[QT][NKT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) s[qt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // ---- S^T = K . Q^T ----
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const int row = k0 + kt * 16 + r;
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(klds + row * (D * 2) + (((ks * 4 + g) ^ (row & KMASK)) << 4));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    s[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ks], s[qt][kt], 0, 0, 0);
            }
        }
        // ---- online softmax per column (lane) ----
        bf16x8 pf[QT][NKT / 2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            // s holds RAW q.k; the softmax scale rides in the exp2 argument (one v_fma per element instead of a
            // multiply and a subtract).  The soft-cap and the visibility mask are whole wave-uniform blocks (one
            // scalar branch each), so full tiles below the causal diagonal run neither.
            float sc = scale_log2e;
            if (capped) {
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // cap * tanh(x / cap) on the natural-log-scale logit, stored in log2 units
                        const float x = s[qt][kt][j] * scale_log2e * 0.6931471805599453f;
                        s[qt][kt][j] = logit_cap * tanhf(x / logit_cap) * 1.4426950408889634f;
                    }
                sc = 1.f;
            }
            if (masked) {
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int key = key_base + k0 + kt * 16 + g * 4 + j;
                        s[qt][kt][j] = key < limit[qt] ? s[qt][kt][j] : -INFINITY;
                    }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) mx = fmaxf(mx, s[qt][kt][j]);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            mx *= sc;                                        // log2 units (sc > 0)
            // Deferred rescale: keep the old reference maximum while the new one is at most 2^thr above it (P stays
            // <= 2^thr, harmless in fp32 sums and in bf16 P); the branch is wave-uniform.  The first tile (m = -inf)
            // and any larger jump take the exact path.
            float m_new = m[qt];
            if (__any(mx > m[qt] + kRescaleThr)) {
                m_new = fmaxf(m[qt], mx);
                const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m[qt] - m_new);
                l[qt] *= alpha;
#pragma unroll
                for (int t = 0; t < VT; ++t) o[qt][t] *= alpha;
                m[qt] = m_new;
            }
            const float neg_m = (m_new == -INFINITY) ? 0.f : -m_new;
            float psum = 0.f;
            float p[NKT][4];
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    p[kt][j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qt][kt][j], sc, neg_m));   // exp2(-inf) = 0 for masked keys
                    psum += p[kt][j];
                }
            l[qt] += psum;     // per-lane partial (this lane's keys); lane groups are summed at the end
#pragma unroll
            for (int ss = 0; ss < NKT / 2; ++ss) {
                u32x4 w;
                w[0] = pack_bf16x2(p[2 * ss][0], p[2 * ss][1]);
                w[1] = pack_bf16x2(p[2 * ss][2], p[2 * ss][3]);
                w[2] = pack_bf16x2(p[2 * ss + 1][0], p[2 * ss + 1][1]);
                w[3] = pack_bf16x2(p[2 * ss + 1][2], p[2 * ss + 1][3]);
                pf[qt][ss] = __builtin_bit_cast(bf16x8, w);
            }
        }
        // ---- O^T += V^T . P ----
        const int q = r >> 2, pp = r & 3;   // transposed read: lane supplies row q, columns 4pp.. of its group's 4x16 block
        if constexpr (ASM_TR) {
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            constexpr int NI = VT * (NKT / 2);
            s16x4 lo[2], hi[2];
            auto rd = [&](int i, s16x4& l, s16x4& h) {
                const int t = i / (NKT / 2), ss = i - t * (NKT / 2);
                const int row0 = k0 + ss * 32 + g * 4 + q, row1 = row0 + 16;
                const int col = dv0 + t * 16 + pp * 4;
                const int ch = col >> 3, sub = (col & 7) * 2;
                const unsigned a0 = (unsigned)(size_t)(lds_s16x4_ptr)(vlds + row0 * (VW * 2) + ((ch ^ (row0 & VMASK)) << 4) + sub);
                const unsigned a1 = (unsigned)(size_t)(lds_s16x4_ptr)(vlds + row1 * (VW * 2) + ((ch ^ (row1 & VMASK)) << 4) + sub);
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(l) : "v"(a0));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(h) : "v"(a1));
            };
            rd(0, lo[0], hi[0]);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                s16x4& l = lo[i & 1];
                s16x4& h = hi[i & 1];
                if (i + 1 < NI) {
                    rd(i + 1, lo[(i + 1) & 1], hi[(i + 1) & 1]);
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(l), "+v"(h));     // all but the two reads just issued
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(l), "+v"(h));
                }
                s16x8 vv;
                vv[0] = l[0]; vv[1] = l[1]; vv[2] = l[2]; vv[3] = l[3];
                vv[4] = h[0]; vv[5] = h[1]; vv[6] = h[2]; vv[7] = h[3];
                const bf16x8 vf = __builtin_bit_cast(bf16x8, vv);
                const int t = i / (NKT / 2), ss = i - t * (NKT / 2);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    o[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][ss], o[qt][t], 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < VT; ++t) {
#pragma unroll
            for (int ss = 0; ss < NKT / 2; ++ss) {
                const int row0 = k0 + ss * 32 + g * 4 + q;
                const int row1 = row0 + 16;
                const int col = dv0 + t * 16 + pp * 4;      // in elements; 4 elements = 8 bytes inside one 16-B chunk
                const int ch = col >> 3, sub = (col & 7) * 2;
                const unsigned char* a0 = vlds + row0 * (VW * 2) + ((ch ^ (row0 & VMASK)) << 4) + sub;
                const unsigned char* a1 = vlds + row1 * (VW * 2) + ((ch ^ (row1 & VMASK)) << 4) + sub;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a1));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                s16x8 vv;
                vv[0] = lo[0]; vv[1] = lo[1]; vv[2] = lo[2]; vv[3] = lo[3];
                vv[4] = hi[0]; vv[5] = hi[1]; vv[6] = hi[2]; vv[7] = hi[3];
                const bf16x8 vf = __builtin_bit_cast(bf16x8, vv);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    o[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][ss], o[qt][t], 0, 0, 0);
            }
        }
    }

    // total of the per-lane partial sums of a column (the 4 lane groups hold disjoint keys)
    SGLK_DEV float column_sum(int qt) const {
        float v = l[qt];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        return v;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// extend attention
// ---------------------------------------------------------------------------------------------------------------------
struct ExtendParams {
    const unsigned short *q, *k_ext, *v_ext, *k_buf, *v_buf;
    unsigned short* o;
    int64_t q_s0, q_s1, ke_s0, ke_s1, ve_s0, ve_s1, kb_s0, kb_s1, vb_s0, vb_s1, o_s0, o_s1;   // element strides [token][head]
    const void* req_to_tokens;   // [B][L]
    int64_t rtt_stride;
    int rtt_is64;
    const int64_t* b_req_idx;
    const int64_t* b_seq_len;
    const int* b_seq_len_extend;
    const int* b_start_loc_extend;
    int HQ, HKV, HBUF;
    float sm_scale, logit_cap;
    // flash_attn_varlen_func form (/root/reference/test_flash_attn_varlen.py:100-108): no paged prefix, queries and keys of
    // sequence b are rows cu_q[b]..cu_q[b+1] of q / cu_k[b]..cu_k[b+1] of k_ext, v_ext; causal = top-left aligned mask
    int varlen, causal;
    const int* cu_q;
    const int* cu_k;
    int d_real, dv_real;     // head dims of the tensors; the kernel's D / DV are these rounded up (zero-padded images)
    int v_aligned;           // V rows / chunks are 16-byte aligned
    int B, nqblk, n_cu, order, pair;   // launch geometry (see the kernel's workgroup-id decomposition)
};

static int attn_cus() { return device_cu_count(); }
static int attn_order(int dflt) {   // A/B override: bit 0 = flip second wave, bit 1 = heads fastest
    return knobs().attn_order >= 0 ? knobs().attn_order : dflt;
}

// heavy + light query blocks in one workgroup (see the kernel): SGLK_ATTN_PAIR = 0 / 1 overrides
static int pair_blocks(int nqblk, int64_t wgs_unpaired, int n_cu, int causal) {
    if (nqblk < 2 || !causal) return 0;
    if (knobs().attn_pair >= 0) return knobs().attn_pair;
    return wgs_unpaired <= 2 * (int64_t)n_cu ? 1 : 0;
}

// FORM (compile time, so that the extend form pays nothing for the others): 0 = extend_attention_cpu (paged prefix +
// causal extend part, exact head dims); 1 = flash_attn_varlen_func with head dims equal to D / DV and 16-byte aligned
// rows; 2 = flash_attn_varlen_func with zero-padded head dims and / or rows that are only 4-byte aligned.
// NW = waves per workgroup: 8 (two per SIMD: one wave's softmax beside the other's MFMAs), or 4 (A/B knob) -- then two
// workgroups share a CU the same way, each streaming its own K/V tiles, with half the queries per workgroup.
template <int D, int DV, int QT, int FORM, int NW = 8>
__global__ __launch_bounds__(NW * 64, 2) void extend_attention_kernel(const ExtendParams p) {
    constexpr bool VARLEN = FORM != 0, RAGGED = FORM == 2;
    // queries per workgroup: NW waves x QT tiles x 16 (QT = 2 where the register budget of 256 per lane allows it)
    constexpr int QB = NW * QT * 16;
    constexpr int WQ = QT * 16;   // queries per wave
    constexpr int KB = kKeys * D * 2, VB = kKeys * DV * 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (KB + VB)];   // two {K, V} tile buffers

    // 1-D grid of nqblk * B * HQ workgroups; the linear id is decomposed here so that the order the hardware dispatches
    // them in can be chosen (p.order bit 1: heads fastest, query blocks slowest = the heaviest query blocks of EVERY
    // (sequence, head) first; else query blocks fastest.  bit 0: the second CU-count's worth of workgroups is taken in
    // reverse, so that the two workgroups resident on a CU at the start are a heavy and a light one).
    // Measured (profiles/r01_v7_attn_order.txt): heads fastest takes causal extend at B=1, ctx=4096, 32 heads from 0.378
    // to 0.229 ms (all 512 workgroups are resident at once, and with query blocks fastest the CUs that drew two heavy
    // blocks set the time); the reversal helps only the query-blocks-fastest order and is off by default.
    int lin = blockIdx.x;
    {
        const int total = gridDim.x, hi = 2 * p.n_cu < total ? 2 * p.n_cu : total;
        if ((p.order & 1) && lin >= p.n_cu && lin < hi) lin = p.n_cu + hi - 1 - lin;
    }
    const int nqblk = p.nqblk;
    // p.pair: a workgroup takes TWO query blocks, the i-th heaviest and the i-th lightest of its (sequence, head), one after the
    // other -- every workgroup of a causal launch then has the same work.  Chosen when the un-paired launch would be resident all
    // at once (<= 2 workgroups per CU), where nothing is dispatched later to even the CUs out.
    const int nitem = p.pair ? (nqblk + 1) / 2 : nqblk;
    int b, h, qblk0;
    if (p.order & 2) { h = lin % p.HQ; b = (lin / p.HQ) % p.B; qblk0 = lin / (p.HQ * p.B); }
    else { qblk0 = lin % nitem; b = (lin / nitem) % p.B; h = lin / (nitem * p.B); }
    int ext_len, prefix, ext_start, k_start = 0, n_keys;
    if (VARLEN) {
        ext_start = p.cu_q[b];
        ext_len = p.cu_q[b + 1] - ext_start;
        k_start = p.cu_k[b];
        n_keys = p.cu_k[b + 1] - k_start;
        prefix = 0;
    } else {
        ext_len = p.b_seq_len_extend[b];
        prefix = (int)p.b_seq_len[b] - ext_len;
        ext_start = p.b_start_loc_extend[b];
        k_start = ext_start;
        n_keys = prefix + ext_len;
    }
    const bool causal = !VARLEN || p.causal;
    const int64_t req = VARLEN ? 0 : p.b_req_idx[b];
    const int kvh = h / (p.HQ / p.HKV);
    const int kvh_buf = p.HBUF == p.HKV ? kvh : 0;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d_real = RAGGED ? p.d_real : D, dv_real = RAGGED ? p.dv_real : DV;
    for (int part = 0; part < (p.pair ? 2 : 1); ++part) {
    const int qblk = part == 0 ? qblk0 : nqblk - 1 - qblk0;
    if (part == 1 && qblk == qblk0) break;
    // heaviest query blocks (most keys under the causal mask) first
    const int q0 = (nqblk - 1 - qblk) * QB;
    if (q0 >= ext_len) continue;
    Core<D, DV, QT> core;
    core.init();
    int limit[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qi = q0 + wave * WQ + qt * 16 + (lane & 15);
        const bool valid = qi < ext_len;
        core.load_q(qt, valid ? p.q + (int64_t)(ext_start + qi) * p.q_s0 + (int64_t)h * p.q_s1 : nullptr, lane, d_real);
        // causal: keys up to and including the query's own position (never past the sequence's keys)
        const int vis = causal ? (prefix + qi + 1 < n_keys ? prefix + qi + 1 : n_keys) : n_keys;
        limit[qt] = valid ? vis : 0;
    }
    KvSource ks, vs;
    const unsigned char* page = reinterpret_cast<const unsigned char*>(p.req_to_tokens) + req * p.rtt_stride * (p.rtt_is64 ? 8 : 4);
    ks.buf = p.k_buf + (int64_t)kvh_buf * p.kb_s1; ks.buf_stride_tok = p.kb_s0; ks.page = page; ks.page_is64 = p.rtt_is64; ks.n_paged = prefix;
    ks.ext = p.k_ext + (int64_t)k_start * p.ke_s0 + (int64_t)kvh * p.ke_s1; ks.ext_stride_tok = p.ke_s0;
    vs = ks;
    vs.buf = p.v_buf + (int64_t)kvh_buf * p.vb_s1; vs.buf_stride_tok = p.vb_s0;
    vs.ext = p.v_ext + (int64_t)k_start * p.ve_s0 + (int64_t)kvh * p.ve_s1; vs.ext_stride_tok = p.ve_s0;

    const int q_last = (q0 + QB < ext_len ? q0 + QB : ext_len);
    // the block's last query sees keys < prefix + q_last (all keys without the causal mask)
    const int kv_end = causal ? (prefix + q_last < n_keys ? prefix + q_last : n_keys) : n_keys;
    const float scale_log2e = p.sm_scale * 1.4426950408889634f;
    const int ntiles = (kv_end + kKeys - 1) / kKeys;

    // software pipeline: tile i+1 travels HBM -> registers while tile i is multiplied out of LDS
    TileRegs<D, NW * 64> kreg;
    TileRegs<DV, NW * 64> vreg;
    // a wave's 32 queries see keys < wave_kv_end: later tiles of the workgroup's stream are fully masked for it
    const int wave_q_last = (q0 + wave * WQ + WQ < ext_len) ? q0 + wave * WQ + WQ : ext_len;
    const int wave_kv_end = causal ? (prefix + wave_q_last < n_keys ? prefix + wave_q_last : n_keys) : n_keys;
    auto nkeys = [&](int t) { const int r = kv_end - t * kKeys; return r < kKeys ? r : kKeys; };
    const bool v_al = !RAGGED || p.v_aligned != 0;
    kreg.load(ks, 0, nkeys(0), d_real, true);
    vreg.load(vs, 0, nkeys(0), dv_real, v_al);
    kreg.store(lds);
    vreg.store(lds + KB);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        unsigned char* cur = lds + (t & 1) * (KB + VB);
        unsigned char* nxt = lds + ((t + 1) & 1) * (KB + VB);
        const bool more = t + 1 < ntiles;
        if (more) {
            kreg.load(ks, (t + 1) * kKeys, nkeys(t + 1), d_real, true);
            vreg.load(vs, (t + 1) * kKeys, nkeys(t + 1), dv_real, v_al);
        }
        if (t * kKeys < wave_kv_end) core.template tile<false>(cur, cur + KB, t * kKeys, limit, scale_log2e, p.logit_cap, lane);
        if (more) {
            kreg.store(nxt);      // `nxt` was last read in iteration t-1, which every wave left before this barrier's
            vreg.store(nxt + KB); // predecessor; the barrier below publishes it for iteration t+1
        }
        __syncthreads();
    }
    // ---- normalise and store: lane (g, column) holds output dims 16t + 4g .. +3 of its query ----
    const int g4 = (lane >> 4) * 4;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float lsum = core.column_sum(qt);
        const int qi = q0 + wave * WQ + qt * 16 + (lane & 15);
        if (qi >= ext_len) continue;
        const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
        unsigned short* orow = p.o + (int64_t)(ext_start + qi) * p.o_s0 + (int64_t)h * p.o_s1;
#pragma unroll
        for (int t = 0; t < DV / 16; ++t) {
            const f32x4 v = core.o[qt][t] * inv;
            uint2 w;
            w.x = pack_bf16x2(v[0], v[1]);
            w.y = pack_bf16x2(v[2], v[3]);
            const int c = t * 16 + g4;                       // output dims c .. c+3
            if (c + 4 <= dv_real && v_al) {
                *reinterpret_cast<uint2*>(orow + c) = w;
            } else {                                          // padded tail / rows that are only 4-byte aligned
                if (c + 2 <= dv_real) *reinterpret_cast<unsigned*>(orow + c) = w.x;
                if (c + 4 <= dv_real) *reinterpret_cast<unsigned*>(orow + c + 2) = w.y;
            }
        }
    }
    }   // part
}

// ---------------------------------------------------------------------------------------------------------------------
// extend attention, two-phase form (D = DV = 128)
// ---------------------------------------------------------------------------------------------------------------------
// The kernel above keeps all eight waves of a workgroup in the same phase of a tile (one barrier per tile): the two waves of a
// SIMD issue their Q.K MFMAs together, then both run the softmax on the VALU, then both issue the P.V MFMAs -- counters of the
// Qwen3 prefill shape: matrix pipe busy 33 % of the cycles, VALU about 45 %, one after the other.  Per tile and wave the softmax
// (32 exp2 at quarter rate plus ~200 other VALU instructions) costs as many cycles as the 64 MFMAs.
// Here the waves form two groups that run HALF A TILE APART: a tile is a VALU slot V_t (softmax of S_t -> P_t) and a matrix
// slot M_t (O += V_t^T P_t, then S_{t+1} = K_{t+1} Q^T), with a workgroup barrier after every slot; group A (waves 0-3) runs
// V_t in slot 2t+1 and M_t in slot 2t+2, group B (waves 4-7, the SECOND wave of each SIMD) one slot later.  In every slot each
// SIMD has one wave on the matrix pipe and one on the VALU.
//   LDS: rings of three K and three V tiles (96 KiB).  M_t reads V_t and K_{t+1} in slots 2t+2 (A) and 2t+3 (B); tile halves
//   K_{t+2} / V_{t+1} are written by each group during ITS V_t slot (2t+1 / 2t+2) into the buffers last read by M_{t-2}
//   (slots 2t-2 / 2t-1), and are complete one barrier before M_{t+1} reads them (2t+4).  Their global loads are issued one
//   slot earlier, so a load has a whole matrix slot to land.
typedef const __attribute__((address_space(1))) void* dma_gptr_t;
typedef __attribute__((address_space(3))) void* dma_lptr_t;

// A group's half of a 64-row tile by LDS-DMA: 256 threads x 2 instructions of 16 bytes per lane, no registers, no ds_write.
// The image is linear per wave (1 KiB per instruction), so the XOR swizzle is applied to the SOURCE chunk a lane fetches.
// Rows past `nkeys` repeat the last real row (finite data: their probabilities are exactly 0).
// V image of the two-phase kernel (16 chunks per row): chunk c of row r sits in slot c ^ vswz(r).  A ds_read_b64_tr_b16 of 32
// lanes takes a 32-byte column pair from 8 consecutive rows; with the K image's 3-bit mask those fall into one 128-byte window
// (half the banks, two passes), with the row number in bits 3:1 they cover all 64 banks.
SGLK_DEV int vswz(int row) { return ((row & 7) << 1) | ((row >> 3) & 1); }

template <int WIDTH, int FORM, bool VIMG = false>
struct HalfDma {
    static constexpr int CH = WIDTH / 8;
    static constexpr int N = CH / 8;                 // instructions per thread: a group's 32 rows x CH chunks over 256 threads
    static_assert(CH % 8 == 0, "row width must be a multiple of 64 elements");
    const unsigned short* g[N];
    SGLK_DEV void prep(const KvSource& src, int p0, int nkeys, int grp, int t256) {
        constexpr int MASK = Swz<CH>::mask;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = grp * (32 * CH) + i * 256 + t256;
            const int row = c / CH, slot = c - row * CH;
            const int rr = row < nkeys ? row : nkeys - 1;
            const int pos = p0 + rr;
            const unsigned short* base;
            if (FORM == 0) {
                const int pi = pos < src.n_paged ? pos : 0;
                int64_t tok = 0;
                if (src.n_paged > 0)
                    tok = src.page_is64 ? reinterpret_cast<const int64_t*>(src.page)[pi] : (int64_t)reinterpret_cast<const int*>(src.page)[pi];
                const unsigned short* a = src.buf + tok * src.buf_stride_tok;
                const unsigned short* b = src.ext + (int64_t)(pos - src.n_paged) * src.ext_stride_tok;
                base = pos < src.n_paged ? a : b;
            } else {
                base = src.ext + (int64_t)pos * src.ext_stride_tok;
            }
            g[i] = base + ((slot ^ (VIMG ? vswz(row) : (row & MASK))) << 3);
        }
    }
    SGLK_DEV void issue(unsigned char* lds, int grp, int wv) const {
#pragma unroll
        for (int i = 0; i < N; ++i)
            __builtin_amdgcn_global_load_lds((dma_gptr_t)g[i], (dma_lptr_t)(lds + (grp * (32 * CH) + i * 256 + wv * 64) * 16), 16, 0, 0);
    }
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
// max over the lanes l ^ 16 / l ^ 32 by row swaps (gfx950 v_permlane16_swap / v_permlane32_swap): with both operands the same
// register the two results hold the two rows of every pair in all lanes of the pair
SGLK_DEV float xor16_max(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}
SGLK_DEV float xor32_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}

#ifndef SGLK_PP_AHEAD
#define SGLK_PP_AHEAD 4
#endif
#ifndef SGLK_PP_AHEAD_192
#define SGLK_PP_AHEAD_192 2
#endif
template <int D_, int AHEAD = SGLK_PP_AHEAD>
struct CorePP {
    static constexpr int D = D_, DV = 128, QT = 2, KS = D_ / 32, VT = 8, NKT = 4;
    static constexpr int kAhead = AHEAD;   // operand fragments (16 bytes per lane) in flight ahead of the MFMAs
    bf16x8 qf[QT][KS];
    f32x4 o[QT][VT];
    f32x4 s[QT][NKT];
    bf16x8 pf[QT][NKT / 2];
    float m[QT], l[QT];

    SGLK_DEV void init() {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            m[qt] = -INFINITY;
            l[qt] = 0.f;
#pragma unroll
            for (int t = 0; t < VT; ++t) o[qt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    SGLK_DEV void load_q(int qt, const unsigned short* qrow, int lane) {
        const int g = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (qrow) v = *reinterpret_cast<const uint4*>(qrow + ks * 32 + g * 8);
            qf[qt][ks] = __builtin_bit_cast(bf16x8, v);
        }
    }
    // S^T = K . Q^T of one 64-key tile.  Only ONE wave of a SIMD is in its matrix slot, so nothing else hides the LDS latency of
    // the operand reads: they are issued eight fragments ahead of the MFMAs that consume them (scheduling fences pin the order;
    // left alone the scheduler keeps one read in flight and every pair of MFMAs waits a full LDS round trip).
    SGLK_DEV void qk(const unsigned char* klds, int lane) {
        constexpr int KMASK = Swz<D / 8>::mask;
        const int r = lane & 15, g = lane >> 4;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) s[qt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 kf[KS * NKT];
#pragma unroll
        for (int i = 0; i < KS * NKT; ++i) {
            const int ks = i / NKT, kt = i % NKT;
            const int row = kt * 16 + r;
#if (SGLK_PP_ABL & 4)       // timing ablation 4: no operand reads
            kf[i] = qf[0][ks];
            (void)row;
#else
            kf[i] = *reinterpret_cast<const bf16x8*>(klds + row * (D * 2) + (((ks * 4 + g) ^ (row & KMASK)) << 4));
#endif
        }
#pragma unroll
        for (int i = 0; i < KS * NKT; ++i) {
            const int ks = i / NKT, kt = i % NKT;
#pragma unroll
#if (SGLK_PP_ABL & 8)       // timing ablation 8: no MFMAs (the reads stay)
            asm volatile("" ::"v"(kf[i]));
            (void)ks; (void)kt;
#else
            for (int qt = 0; qt < QT; ++qt)
                s[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i], qf[qt][ks], s[qt][kt], 0, 0, 0);
#endif
        }
        __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
        for (int i = 0; i < KS * NKT - kAhead; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < kAhead; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
    // online softmax of the tile in `s` -> `pf` (same arithmetic as Core::tile)
    SGLK_DEV void softmax(int key_base, const int (&limit)[QT], float scale_log2e, float logit_cap, int lane) {
#if (SGLK_PP_ABL & 1)     // timing ablation: no softmax arithmetic (wrong results)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int ss = 0; ss < NKT / 2; ++ss) {
                u32x4 w;
                w[0] = pack_bf16x2(s[qt][2 * ss][0], s[qt][2 * ss][1]);
                w[1] = pack_bf16x2(s[qt][2 * ss][2], s[qt][2 * ss][3]);
                w[2] = pack_bf16x2(s[qt][2 * ss + 1][0], s[qt][2 * ss + 1][1]);
                w[3] = pack_bf16x2(s[qt][2 * ss + 1][2], s[qt][2 * ss + 1][3]);
                pf[qt][ss] = __builtin_bit_cast(bf16x8, w);
            }
        return;
#endif
        int lim_min = limit[0] < limit[1] ? limit[0] : limit[1];
        const bool masked = __any(key_base + NKT * 16 > lim_min);
        const bool capped = __builtin_amdgcn_readfirstlane(logit_cap > 0.f);
        constexpr float kRescaleThr = 8.0f;
        const int g = lane >> 4;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float sc = scale_log2e;
            if (capped) {
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float x = s[qt][kt][j] * scale_log2e * 0.6931471805599453f;
                        s[qt][kt][j] = logit_cap * tanhf(x / logit_cap) * 1.4426950408889634f;
                    }
                sc = 1.f;
            }
            if (masked) {
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int key = key_base + kt * 16 + g * 4 + j;
                        s[qt][kt][j] = key < limit[qt] ? s[qt][kt][j] : -INFINITY;
                    }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) mx = fmaxf(mx, s[qt][kt][j]);
            // the column's four lane groups: row swaps on the VALU (no LDS round trip -- this wave is alone on its SIMD's VALU
            // during the slot, nothing would hide a ds_bpermute's latency)
            mx = xor16_max(mx);
            mx = xor32_max(mx);
            mx *= sc;
            float m_new = m[qt];
            if (__any(mx > m[qt] + kRescaleThr)) {
                m_new = fmaxf(m[qt], mx);
                const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m[qt] - m_new);
                l[qt] *= alpha;
#pragma unroll
                for (int t = 0; t < VT; ++t) o[qt][t] *= alpha;
                m[qt] = m_new;
            }
            const float neg_m = (m_new == -INFINITY) ? 0.f : -m_new;
            // packed fp32 (v_pk_fma_f32 / v_pk_add_f32: two elements per instruction)
            const f32x2 sc2 = {sc, sc}, nm2 = {neg_m, neg_m};
            f32x2 ps = {0.f, 0.f};
            float pr[NKT][4];
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const f32x2 lo = {s[qt][kt][0], s[qt][kt][1]}, hi = {s[qt][kt][2], s[qt][kt][3]};
                const f32x2 y0 = __builtin_elementwise_fma(lo, sc2, nm2), y1 = __builtin_elementwise_fma(hi, sc2, nm2);
                const f32x2 e0 = {__builtin_amdgcn_exp2f(y0[0]), __builtin_amdgcn_exp2f(y0[1])};   // exp2(-inf) = 0 for masked keys
                const f32x2 e1 = {__builtin_amdgcn_exp2f(y1[0]), __builtin_amdgcn_exp2f(y1[1])};
                ps += e0;
                ps += e1;
                pr[kt][0] = e0[0]; pr[kt][1] = e0[1]; pr[kt][2] = e1[0]; pr[kt][3] = e1[1];
            }
            l[qt] += ps[0] + ps[1];
#pragma unroll
            for (int ss = 0; ss < NKT / 2; ++ss) {
                u32x4 w;
                w[0] = pack_bf16x2(pr[2 * ss][0], pr[2 * ss][1]);
                w[1] = pack_bf16x2(pr[2 * ss][2], pr[2 * ss][3]);
                w[2] = pack_bf16x2(pr[2 * ss + 1][0], pr[2 * ss + 1][1]);
                w[3] = pack_bf16x2(pr[2 * ss + 1][2], pr[2 * ss + 1][3]);
                pf[qt][ss] = __builtin_bit_cast(bf16x8, w);
            }
        }
    }
    // O^T += V^T . P  (operand reads eight ahead, as in qk)
    SGLK_DEV void pv(const unsigned char* vlds, int lane) {
        const int r = lane & 15, g = lane >> 4;
        const int q = r >> 2, pp = r & 3;
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x4 lo[VT * 2], hi[VT * 2];
#pragma unroll
        for (int i = 0; i < VT * 2; ++i) {
            const int t = i >> 1, ss = i & 1;
            const int row0 = ss * 32 + g * 4 + q;
            const int row1 = row0 + 16;
            const int col = t * 16 + pp * 4;
            const int ch = col >> 3, sub = (col & 7) * 2;
            const unsigned char* a0 = vlds + row0 * (DV * 2) + ((ch ^ vswz(row0)) << 4) + sub;
            const unsigned char* a1 = vlds + row1 * (DV * 2) + ((ch ^ vswz(row1)) << 4) + sub;
#if (SGLK_PP_ABL & 4)
            lo[i] = __builtin_bit_cast(s16x4, (uint2){(unsigned)i, 0u});
            hi[i] = lo[i];
            (void)a0; (void)a1;
#else
            lo[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));
            hi[i] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a1));
#endif
        }
#pragma unroll
        for (int i = 0; i < VT * 2; ++i) {
            const int t = i >> 1, ss = i & 1;
            s16x8 vv;
            vv[0] = lo[i][0]; vv[1] = lo[i][1]; vv[2] = lo[i][2]; vv[3] = lo[i][3];
            vv[4] = hi[i][0]; vv[5] = hi[i][1]; vv[6] = hi[i][2]; vv[7] = hi[i][3];
            const bf16x8 vf = __builtin_bit_cast(bf16x8, vv);
#if (SGLK_PP_ABL & 8)
            asm volatile("" ::"v"(vf));
            (void)t; (void)ss;
#else
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
                o[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][ss], o[qt][t], 0, 0, 0);
#endif
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * kAhead, 0);
#pragma unroll
        for (int i = 0; i < 16 - kAhead; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
#pragma unroll
        for (int i = 0; i < kAhead; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
    SGLK_DEV float column_sum(int qt) const {
        float v = l[qt];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        return v;
    }
};

// end of a slot: LDS writes of this wave have landed (lgkmcnt(0)); the staged global loads stay in flight across the barrier
// (__syncthreads would wait for them -- a slot would then last at least one memory round trip)
SGLK_DEV void slot_barrier_m() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
SGLK_DEV void slot_barrier_v() {             // ... and this wave's LDS-DMA halves have landed
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0070);     // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

#ifndef SGLK_PP_PRIO
#define SGLK_PP_PRIO 2
#endif
// D = 128 or 192 (the MLA prefill head of /root/reference/bench_extend.py:111-112), DV = 128
template <int FORM, int D = 128>   // FORM 0 = extend_attention_cpu, 1 = flash_attn_varlen_func (exact head dims, aligned rows)
__global__ __launch_bounds__(512, 2) void extend_pp_kernel(const ExtendParams p) {
    constexpr bool VARLEN = FORM != 0;
    constexpr int DV = 128, QT = 2, QB = 256, WQ = 32;
    constexpr int KB = kKeys * D * 2, VB = kKeys * DV * 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char pp_lds[];   // K ring [3][KB], V ring [3][VB]
    unsigned char* const kring = pp_lds;
    unsigned char* const vring = pp_lds + 3 * KB;

    const int lin = blockIdx.x;
    const int nqblk = p.nqblk;
    int b, h, qblk;
    if (p.order & 2) { h = lin % p.HQ; b = (lin / p.HQ) % p.B; qblk = lin / (p.HQ * p.B); }
    else { qblk = lin % nqblk; b = (lin / nqblk) % p.B; h = lin / (nqblk * p.B); }
    int ext_len, prefix, ext_start, k_start = 0, n_keys;
    if (VARLEN) {
        ext_start = p.cu_q[b];
        ext_len = p.cu_q[b + 1] - ext_start;
        k_start = p.cu_k[b];
        n_keys = p.cu_k[b + 1] - k_start;
        prefix = 0;
    } else {
        ext_len = p.b_seq_len_extend[b];
        prefix = (int)p.b_seq_len[b] - ext_len;
        ext_start = p.b_start_loc_extend[b];
        k_start = ext_start;
        n_keys = prefix + ext_len;
    }
    const bool causal = !VARLEN || p.causal;
    const int q0 = (nqblk - 1 - qblk) * QB;      // heaviest query blocks first
    if (q0 >= ext_len) return;
    const int64_t req = VARLEN ? 0 : p.b_req_idx[b];
    const int kvh = h / (p.HQ / p.HKV);
    const int kvh_buf = p.HBUF == p.HKV ? kvh : 0;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = wave >> 2;                    // 0: group A, 1: group B (one slot behind)
    // the two waves of a SIMD take neighbouring 32-query slices, so that they see about the same number of tiles
    const int qslot = ((wave & 3) << 1) | grp;
    CorePP<D, D == 128 ? SGLK_PP_AHEAD : SGLK_PP_AHEAD_192> core;
    core.init();
    int limit[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qi = q0 + qslot * WQ + qt * 16 + (lane & 15);
        const bool valid = qi < ext_len;
        core.load_q(qt, valid ? p.q + (int64_t)(ext_start + qi) * p.q_s0 + (int64_t)h * p.q_s1 : nullptr, lane);
        const int vis = causal ? (prefix + qi + 1 < n_keys ? prefix + qi + 1 : n_keys) : n_keys;
        limit[qt] = valid ? vis : 0;
    }
    KvSource ks, vs;
    const unsigned char* page = reinterpret_cast<const unsigned char*>(p.req_to_tokens) + req * p.rtt_stride * (p.rtt_is64 ? 8 : 4);
    ks.buf = p.k_buf + (int64_t)kvh_buf * p.kb_s1; ks.buf_stride_tok = p.kb_s0; ks.page = page; ks.page_is64 = p.rtt_is64; ks.n_paged = prefix;
    ks.ext = p.k_ext + (int64_t)k_start * p.ke_s0 + (int64_t)kvh * p.ke_s1; ks.ext_stride_tok = p.ke_s0;
    vs = ks;
    vs.buf = p.v_buf + (int64_t)kvh_buf * p.vb_s1; vs.buf_stride_tok = p.vb_s0;
    vs.ext = p.v_ext + (int64_t)k_start * p.ve_s0 + (int64_t)kvh * p.ve_s1; vs.ext_stride_tok = p.ve_s0;

    const int q_last = (q0 + QB < ext_len ? q0 + QB : ext_len);
    const int kv_end = causal ? (prefix + q_last < n_keys ? prefix + q_last : n_keys) : n_keys;
    const float scale_log2e = p.sm_scale * 1.4426950408889634f;
    const int nt = (kv_end + kKeys - 1) / kKeys;            // tiles of the workgroup (>= 1)
    const int wave_q_last = (q0 + qslot * WQ + WQ < ext_len) ? q0 + qslot * WQ + WQ : ext_len;
    const int wave_kv_end = causal ? (prefix + wave_q_last < n_keys ? prefix + wave_q_last : n_keys) : n_keys;
    const int nt_w = q0 + qslot * WQ < ext_len ? (wave_kv_end + kKeys - 1) / kKeys : 0;   // tiles this wave's queries can see
    auto nkeys = [&](int t) { const int r = kv_end - t * kKeys; return r < kKeys ? r : kKeys; };

    // ---- prologue: K_0, K_1, V_0 by all threads (two halves per tile and thread group) ----
    const int t256 = threadIdx.x & 255, wv = t256 >> 6;
    HalfDma<D, FORM> kd;
    HalfDma<DV, FORM, true> vd;
    kd.prep(ks, 0, nkeys(0), grp, t256);
    kd.issue(kring, grp, wv);
    vd.prep(vs, 0, nkeys(0), grp, t256);
    vd.issue(vring, grp, wv);
    if (nt > 1) {
        kd.prep(ks, kKeys, nkeys(1), grp, t256);
        kd.issue(kring + KB, grp, wv);
    }
    // pointers of the halves that travel during V_t: K_{t+2}, V_{t+1}  (page lookups one slot before the DMA instructions: loads
    // retire in order, a lookup issued behind DMA instructions could not be consumed before those have landed)
    auto stage_prep = [&](int t) __attribute__((always_inline)) {
        if (t + 2 < nt) kd.prep(ks, (t + 2) * kKeys, nkeys(t + 2), grp, t256);
        if (t + 1 < nt) vd.prep(vs, (t + 1) * kKeys, nkeys(t + 1), grp, t256);
    };
    auto stage_issue = [&](int t) __attribute__((always_inline)) {
        if (t + 2 < nt) kd.issue(kring + ((t + 2) % 3) * KB, grp, wv);
        if (t + 1 < nt) vd.issue(vring + ((t + 1) % 3) * VB, grp, wv);
    };
    stage_prep(0);
    slot_barrier_v();
    if (grp == 1) slot_barrier_m();             // slot 0: group B idles
    // slot: S_0
    if (nt_w > 0) core.qk(kring, lane);
    stage_issue(0);
    slot_barrier_m();
    for (int t = 0; t < nt; ++t) {
        // ---- V_t ----
        stage_prep(t + 1);
        if (t < nt_w) core.softmax(t * kKeys, limit, scale_log2e, p.logit_cap, lane);
        slot_barrier_v();                       // the halves issued one slot ago have landed
        // ---- M_t ----
        __builtin_amdgcn_s_setprio(SGLK_PP_PRIO);    // the matrix-slot wave wins the SIMD's issue arbitration (+2 %)
#if !(SGLK_PP_ABL & 2)   // timing ablation 2: no matrix slot (wrong results)
        if (t < nt_w) core.pv(vring + (t % 3) * VB, lane);
        if (t + 1 < nt_w) core.qk(kring + ((t + 1) % 3) * KB, lane);
#endif
        __builtin_amdgcn_s_setprio(0);
        stage_issue(t + 1);                     // behind the slot's last LDS read
        slot_barrier_m();
    }
    if (grp == 0) slot_barrier_m();             // last slot: group A idles
    // ---- normalise and store ----
    const int g4 = (lane >> 4) * 4;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float lsum = core.column_sum(qt);
        const int qi = q0 + qslot * WQ + qt * 16 + (lane & 15);
        if (qi >= ext_len) continue;
        const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
        unsigned short* orow = p.o + (int64_t)(ext_start + qi) * p.o_s0 + (int64_t)h * p.o_s1;
#pragma unroll
        for (int t = 0; t < DV / 16; ++t) {
            const f32x4 v = core.o[qt][t] * inv;
            uint2 w;
            w.x = pack_bf16x2(v[0], v[1]);
            w.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(orow + t * 16 + g4) = w;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// decode attention: cache write, split-KV partials, merge
// ---------------------------------------------------------------------------------------------------------------------
// k_buffer[loc[b]] = key[b]; v_buffer[loc[b]] = value[b]  (bit-exact copies; /root/reference/test_mla.py:27,174-175)
__global__ __launch_bounds__(256) void kv_cache_write_kernel(unsigned short* k_buf, int64_t kb_s0, int64_t kb_s1,
                                                             unsigned short* v_buf, int64_t vb_s0, int64_t vb_s1,
                                                             const unsigned short* key, int64_t k_s0, int64_t k_s1,
                                                             const unsigned short* value, int64_t v_s0, int64_t v_s1,
                                                             const void* loc, int loc_is64, int B, int HKV, int D, int DV) {
    const int b = blockIdx.x, h = blockIdx.y;
    const int64_t tok = loc_is64 ? reinterpret_cast<const int64_t*>(loc)[b] : (int64_t)reinterpret_cast<const int*>(loc)[b];
    for (int c = threadIdx.x; c < D; c += 256) k_buf[tok * kb_s0 + h * kb_s1 + c] = key[(int64_t)b * k_s0 + h * k_s1 + c];
    for (int c = threadIdx.x; c < DV; c += 256) v_buf[tok * vb_s0 + h * vb_s1 + c] = value[(int64_t)b * v_s0 + h * v_s1 + c];
}

// A/B knobs of the decode kernel (compile time): images up to this size are double buffered; wave groups for wide heads
#ifndef SGLK_DEC_DOUBLE_LIMIT
#define SGLK_DEC_DOUBLE_LIMIT (150 * 1024)
#endif
#ifndef SGLK_DEC_KH_WIDE
#define SGLK_DEC_KH_WIDE 2
#endif

struct DecodeParams {
    const unsigned short *q, *k_buf, *v_buf;
    int64_t q_s0, q_s1, kb_s0, kb_s1, vb_s0, vb_s1;
    float* logits;          // [B][HQ][splits][DV+1]
    const void* req_to_token;
    int64_t rtt_stride;
    int rtt_is64;
    const int64_t* b_req_idx;
    const int64_t* b_seq_len;
    int HQ, HKV, splits, logit_splits, v_alias;   // splits used / depth of the caller's scratch
    float sm_scale, logit_cap;
    int nt;                 // cache rows are read non-temporal
    // direct: one split per request -> the kernel rounds and stores the output itself and no merge is launched (what the merge of ONE
    // split computes is bf16(1.0 * v / 1.0): the same bits)
    unsigned short* o;
    int64_t o_s0, o_s1;
    int direct;
    // fold_write: the new token's rows (k_buffer[loc[b]] = key[b], v_buffer[loc[b]] = value[b]) are written by the attention
    // kernel itself -- EVERY workgroup writes the rows of its kv head for all B requests before it requests anything, so whatever
    // cache row a page table names (another request's new row included) holds the new bytes by the time it is read
    const unsigned short *key, *value;
    const void* loc;
    int64_t key_s0, key_s1, val_s0, val_s1;
    int loc_is64, B, fold_write;
    int abl;                // developer build only (SGLK_ABL): 1 = no tile arithmetic, 2 = no row requests after the prologue
};

// LDS-DMA staging of one 64-key tile (no registers, asynchronous): the image is written linearly, 1 KiB per wave
// instruction, and the 16-byte-chunk swizzle is applied to the SOURCE address.  Rows past `nkeys` re-read the last valid
// row (finite data: their probabilities are exactly 0, but 0 * garbage must not be NaN).  Returns nothing to wait on:
// the caller counts the instructions (kDmaPerThread) in its s_waitcnt vmcnt.
template <int WIDTH, int THREADS>   // THREADS = the decode kernel's workgroup size (256 or 512)
struct TileDma {
    static constexpr int CH = WIDTH / 8;
    static constexpr int N = (kKeys * CH + THREADS - 1) / THREADS;   // DMA instructions per wave per tile
    static_assert((kKeys * CH) % THREADS == 0, "the image must be a whole number of workgroup-wide DMA rounds");
    // Decode kernel: the cache rows (page-table entries) of a tile are themselves fetched by LDS-DMA -- ONE 256-byte instruction of
    // one wave per tile, into a small ring of id tables -- a whole step before the tile's rows are requested, and the row
    // requests read them from LDS.  (Per-lane lookups cost N loads and N registers per lane and tile; and loads retire in order,
    // so a lookup queued behind the previous tile's rows could not be consumed before those had landed.)  Positions past `nkeys`
    // repeat the last valid row.  An int64 table contributes its low words (cache rows are < 2^31).
    SGLK_DEV static void lookup_dma(unsigned* ids, const KvSource& src, int p0, int nkeys, int lane) {
        const int rr = lane < nkeys ? lane : nkeys - 1;
        const unsigned char* g = reinterpret_cast<const unsigned char*>(src.page) + (int64_t)(p0 + rr) * (src.page_is64 ? 8 : 4);
        __builtin_amdgcn_global_load_lds((dma_gptr_t)g, (dma_lptr_t)ids, 4, 0, 0);
    }
    template <int AUX>   // 2 = non-temporal: rows that one workgroup reads once need not stay in L2 / the Infinity Cache
    SGLK_DEV static void issue_rows(unsigned char* lds, const KvSource& src, const unsigned* ids, int wave, int lane) {
        constexpr int MASK = Swz<CH>::mask;
        int tok[N];
#pragma unroll
        for (int i = 0; i < N; ++i) tok[i] = (int)ids[(i * THREADS + wave * 64 + lane) / CH];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = i * THREADS + wave * 64 + lane;
            const int row = c / CH, slot = c - row * CH;
            const unsigned short* g = src.buf + (int64_t)tok[i] * src.buf_stride_tok + ((slot ^ (row & MASK)) << 3);
            __builtin_amdgcn_global_load_lds((dma_gptr_t)g, (dma_lptr_t)(lds + (i * THREADS + wave * 64) * 16), 16, 0, AUX);
        }
    }
    SGLK_DEV static void issue(unsigned char* lds, const KvSource& src, int p0, int nkeys, int wave, int lane) {
        constexpr int MASK = Swz<CH>::mask;
        constexpr int HALF = (N + 1) / 2;
        // two passes per half: all row lookups first (independent loads, one wait), then the DMA instructions
#pragma unroll
        for (int h0 = 0; h0 < N; h0 += HALF) {
            const unsigned short* g[HALF];
#pragma unroll
            for (int j = 0; j < HALF; ++j) {
                const int i = h0 + j;
                if (i < N) {
                    const int c = i * THREADS + wave * 64 + lane;        // linear chunk of the image this lane fills
                    const int row = c / CH, slot = c - row * CH;
                    const int rr = row < nkeys ? row : nkeys - 1;
                    g[j] = kv_row_paged(src, p0 + rr) + ((slot ^ (row & MASK)) << 3);
                }
            }
#pragma unroll
            for (int j = 0; j < HALF; ++j) {
                const int i = h0 + j;
                if (i < N)
                    __builtin_amdgcn_global_load_lds((dma_gptr_t)g[j], (dma_lptr_t)(lds + (i * THREADS + wave * 64) * 16), 16, 0, 0);
            }
        }
    }
};

// 8 waves = 2 key halves x CT column tiles (16 q heads each) x NDV slices of the value width (CT * NDV = 4): a wave computes
// the logits of its 16 heads against ITS 32 keys of every 64-key tile and accumulates DV / NDV output columns over them --
// an independent online-softmax state per wave; the two key halves of a (column tile, slice) are merged once, through LDS,
// after the last tile.  (With 4 waves each taking all 64 keys the logits were computed NDV times over and one wave per SIMD
// had nothing to hide its LDS reads and DMA issue behind.)  The accumulator of the widest case (MLA, DV = 512, 22 heads ->
// CT = 2, NDV = 2) is 64 registers.  K/V tiles arrive by LDS-DMA, two tiles deep when the images fit (tile t+1 flies while
// tile t is multiplied).
// KH = 1: 4 waves, each takes all 64 keys of a tile (no merge) -- faster for narrow heads (D = 128: 4.6 vs 4.4 TB/s), where
// the logits are a small part of the work and two workgroups share a CU anyway.
template <int D, int DV, bool V_ALIAS, int NDV, int KH>
__global__ __launch_bounds__(256 * KH, 1) void decode_attention_kernel(const DecodeParams p) {
    typedef TileDma<D, 256 * KH> DmaK;
    typedef TileDma<DV, 256 * KH> DmaV;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    constexpr int KB = kKeys * D * 2, VB = V_ALIAS ? 0 : kKeys * DV * 2;
    constexpr bool kDouble = 2 * (KB + VB) <= SGLK_DEC_DOUBLE_LIMIT;
    constexpr int DVW = DV / NDV;
    constexpr int kDma = DmaK::N + (V_ALIAS ? 0 : DmaV::N);

    const int b = blockIdx.x, kvh = blockIdx.y, split = blockIdx.z;
    const int group = p.HQ / p.HKV;
    const int seq_len = (int)p.b_seq_len[b];
    const int per = (seq_len + p.splits - 1) / p.splits;
    const int k_begin = split * per;
    const int k_end = (k_begin + per < seq_len) ? k_begin + per : seq_len;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kh = KH == 2 ? wave >> 2 : 0, w4 = wave & 3;   // key half; (column tile, value slice) index
    const int ct = w4 / NDV, dsl = w4 - ct * NDV;       // column tile, value slice
    const int col = ct * 16 + (lane & 15);              // q head inside the group handled by this lane's column
    const bool col_valid = col < group;
    const bool wave_active = ct * 16 < group;
    const int h = kvh * group + col;

    Core<D, DVW, 1> core;
    core.init();
    if (wave_active) core.load_q(0, col_valid ? p.q + (int64_t)b * p.q_s0 + (int64_t)h * p.q_s1 : nullptr, lane);
    int limit[1] = {col_valid ? k_end : 0};

    const int64_t req = p.b_req_idx[b];
    KvSource ks, vs;
    const unsigned char* page = reinterpret_cast<const unsigned char*>(p.req_to_token) + req * p.rtt_stride * (p.rtt_is64 ? 8 : 4);
    ks.buf = p.k_buf + (int64_t)kvh * p.kb_s1; ks.buf_stride_tok = p.kb_s0; ks.page = page; ks.page_is64 = p.rtt_is64;
    ks.n_paged = seq_len; ks.ext = nullptr; ks.ext_stride_tok = 0;
    vs = ks;
    vs.buf = p.v_buf + (int64_t)kvh * p.vb_s1; vs.buf_stride_tok = p.vb_s0;
    const float scale_log2e = p.sm_scale * 1.4426950408889634f;

    const int ntiles = (k_end - k_begin + kKeys - 1) / kKeys;
    auto issue = [&](int t, unsigned char* buf) {
        const int p0 = k_begin + t * kKeys;
        const int nk = k_end - p0 < kKeys ? k_end - p0 : kKeys;
        DmaK::issue(buf, ks, p0, nk, wave, lane);
        if (!V_ALIAS) DmaV::issue(buf + KB, vs, p0, nk, wave, lane);
    };
    // Double-buffered form.  Tile t+2's rows are requested the moment every wave has left tile t's buffer -- BEFORE the wait
    // for tile t+1 -- so that one to two tiles are in flight at all times (with the request placed behind that wait the CU
    // had only tile t+1 in flight, while it multiplied and while it stalled: a tile cost a whole memory round trip, 4.6 TB/s
    // at best).  The id table of tile t+3 is requested just before those rows (TileDma::lookup_dma), i.e. it has a whole step
    // to arrive and never queues in front of rows that wait for it.
    // Queue of a wave at the wait of tile t:  ... rows(t) | [ids(t+2)] rows(t+1)  -> s_waitcnt vmcnt(kDma) covers rows(t), ids(t+2).
    unsigned* const idring = reinterpret_cast<unsigned*>(dyn_lds + (kDouble ? 2 : 1) * (KB + VB));   // 4 tables of 64 ids
    auto lookup = [&](int t) {
        if (wave != 0) return;
        const int tt = t < ntiles ? t : ntiles - 1;          // past the end: the last tile's ids again (never used)
        const int p0 = k_begin + tt * kKeys;
        const int nk = k_end - p0 < kKeys ? k_end - p0 : kKeys;
        DmaK::lookup_dma(idring + (t & 3) * kKeys, ks, p0, nk, lane);
    };
    auto issue_rows = [&](int t, unsigned char* buf) {
        const unsigned* ids = idring + (t & 3) * kKeys;
        if (p.nt) {
            DmaK::template issue_rows<2>(buf, ks, ids, wave, lane);
            if constexpr (!V_ALIAS) DmaV::template issue_rows<2>(buf + KB, vs, ids, wave, lane);
        } else {
            DmaK::template issue_rows<0>(buf, ks, ids, wave, lane);
            if constexpr (!V_ALIAS) DmaV::template issue_rows<0>(buf + KB, vs, ids, wave, lane);
        }
    };
    auto compute = [&](int t, const unsigned char* cur) {
        if (wave_active && !SGLK_ABL(p.abl, 1))
            core.template tile<V_ALIAS, DV, 4 / KH, true>(cur, V_ALIAS ? cur : cur + KB, k_begin + t * kKeys, limit, scale_log2e,
                                                          p.logit_cap, lane, dsl * DVW, kh * 32);
    };
    if constexpr (kDouble) {
        static_assert(kDma <= 63, "the counted wait must fit s_waitcnt's vmcnt field");
        unsigned char* const buf0 = dyn_lds;
        unsigned char* const buf1 = dyn_lds + KB + VB;
        if (ntiles > 0) {
            lookup(0);
            lookup(1);
            lookup(2);
        }
        if (p.fold_write) {
            // 16-byte chunks; V after K in program order like the reference's two assignments (MLA: v_buffer aliases k_buffer's
            // first DV columns, and `value` wins there).  Identical bytes from every workgroup: the race is benign, and a
            // workgroup's own stores have reached its XCD's L2 (vmcnt) before any wave of it reads a row.
            constexpr int KC = D / 8, VC = DV / 8;
            for (int i = threadIdx.x; i < p.B * (KC + VC); i += 256 * KH) {
                const int b2 = i / (KC + VC), c = i - b2 * (KC + VC);
                const int64_t tok = p.loc_is64 ? reinterpret_cast<const int64_t*>(p.loc)[b2] : (int64_t)reinterpret_cast<const int*>(p.loc)[b2];
                if (c < KC) {
                    if (V_ALIAS && c < VC) continue;              // overwritten by the value chunk below
                    const uint4 v = *reinterpret_cast<const uint4*>(p.key + (int64_t)b2 * p.key_s0 + (int64_t)kvh * p.key_s1 + c * 8);
                    *reinterpret_cast<uint4*>(const_cast<unsigned short*>(p.k_buf) + tok * p.kb_s0 + (int64_t)kvh * p.kb_s1 + c * 8) = v;
                } else {
                    const int cv = c - KC;
                    const uint4 v = *reinterpret_cast<const uint4*>(p.value + (int64_t)b2 * p.val_s0 + (int64_t)kvh * p.val_s1 + cv * 8);
                    *reinterpret_cast<uint4*>(const_cast<unsigned short*>(p.v_buf) + tok * p.vb_s0 + (int64_t)kvh * p.vb_s1 + cv * 8) = v;
                }
            }
        }
        if (ntiles > 0 || p.fold_write) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (ntiles > 0) {
            issue_rows(0, buf0);
            if (ntiles > 1) issue_rows(1, buf1);
        }
        for (int t = 0; t < ntiles; ++t) {
            unsigned char* const cur = (t & 1) ? buf1 : buf0;
            if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // bare s_barrier: __syncthreads()' fence would make the compiler drain vmcnt -- the rows in flight -- in front of it
            __builtin_amdgcn_s_barrier();         // every wave's share of tile t has landed, and the id table of tile t+2
            asm volatile("" ::: "memory");
            compute(t, cur);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();         // every wave has left `cur`
            asm volatile("" ::: "memory");
            if (t + 2 < ntiles) {
                lookup(t + 3);
                if (!SGLK_ABL(p.abl, 2)) issue_rows(t + 2, cur);
            }
        }
    } else {
        if (ntiles > 0) issue(0, dyn_lds);
        for (int t = 0; t < ntiles; ++t) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(t, dyn_lds);
            __syncthreads();                     // the single buffer is free again
            if (t + 1 < ntiles) issue(t + 1, dyn_lds);
        }
    }
    (void)kDma;
    // ---- merge the two key halves: the upper half parks {o, m, l} lane for lane in LDS (the tile buffers are dead), the
    //      lower half rescales both to the common maximum and adds ----
    float lsum = core.column_sum(0);
    constexpr int XS = DVW / 4 + 2;                     // floats per lane: its o registers, m, l
    float* xch = reinterpret_cast<float*>(dyn_lds) + (size_t)(w4 * 64 + lane) * XS;
    if (KH == 2) __syncthreads();
    if (KH == 2 && kh == 1 && wave_active) {
#pragma unroll
        for (int t = 0; t < DVW / 16; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) xch[t * 4 + j] = core.o[0][t][j];
        xch[DVW / 4] = core.m[0];
        xch[DVW / 4 + 1] = lsum;
    }
    if (KH == 2) __syncthreads();
    if (kh == 1 || !wave_active) return;
    if (KH == 2) {
        const float m1 = xch[DVW / 4], l1 = xch[DVW / 4 + 1];
        const float m0 = core.m[0];
        const float mt = fmaxf(m0, m1);
        // a half that saw no visible key has m = -inf and contributes nothing (and exp2(-inf - -inf) must not be taken)
        const float a0 = (m0 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m0 - mt);
        const float a1 = (m1 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m1 - mt);
#pragma unroll
        for (int t = 0; t < DVW / 16; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) core.o[0][t][j] = core.o[0][t][j] * a0 + xch[t * 4 + j] * a1;
        lsum = lsum * a0 + l1 * a1;
        core.m[0] = mt;
    }
    if (!col_valid) return;
    float* dst = p.logits + (((int64_t)b * p.HQ + h) * p.logit_splits + split) * (DV + 1);
    const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
    const int g4 = (lane >> 4) * 4;
    if (p.direct) {
        unsigned short* orow = p.o + (int64_t)b * p.o_s0 + (int64_t)h * p.o_s1 + dsl * DVW + g4;
#pragma unroll
        for (int t = 0; t < DVW / 16; ++t) {
            const f32x4 v = core.o[0][t] * inv;
            uint2 w;
            w.x = pack_bf16x2(v[0], v[1]);
            w.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(orow + t * 16) = w;
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < DVW / 16; ++t) {
        const f32x4 v = core.o[0][t] * inv;
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[dsl * DVW + t * 16 + g4 + j] = v[j];
    }
    // log-sum-exp of the split in log2 units (m is in log2 units); -inf marks an empty split
    if (dsl == 0 && (lane >> 4) == 0) dst[DV] = lsum > 0.f ? core.m[0] + __builtin_amdgcn_logf(lsum) : -INFINITY;
}

__global__ __launch_bounds__(256) void decode_merge_kernel(const float* __restrict__ logits, unsigned short* __restrict__ o,
                                                           int64_t o_s0, int64_t o_s1, int HQ, int splits, int logit_splits, int DV) {
    // the splits' log-sum-exps are read ONCE (a lane each) and turned into weights in LDS; every thread then streams its columns
    // with all splits' loads in flight (three dependent passes over the lse values per thread made this kernel 7.6 us at B = 40)
    __shared__ float wgt[64];
    __shared__ float wsum_s;
    const int b = blockIdx.x, h = blockIdx.y;
    const float* src = logits + ((int64_t)b * HQ + h) * logit_splits * (DV + 1);
    // the partial rows do not depend on the weights: the first eight splits' values of this thread's (at most two, DV <= 512)
    // columns are requested BEFORE the barrier, so the lse round trip and the partials' round trip overlap instead of following each
    // other (a split that holds no key has lse = -inf and possibly stale numbers in its row: selected away, never multiplied)
    constexpr int kPre = 8;
    float pre[2][kPre];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = threadIdx.x + it * 256;
#pragma unroll
        for (int s = 0; s < kPre; ++s) pre[it][s] = (c < DV && s < splits) ? src[s * (DV + 1) + c] : 0.f;
    }
    if (threadIdx.x < 64) {
        const int s = threadIdx.x;
        const float lse = s < splits ? src[s * (DV + 1) + DV] : -INFINITY;
        float mx = lse;
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o2));
        const float w = (lse == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(lse - mx);
        wgt[s] = w;
        // fp32 sum in ascending split order, like the serial loop it replaces
        float tot = 0.f;
        for (int k = 0; k < splits; ++k) tot += __shfl(w, k);
        if (s == 0) wsum_s = tot;
    }
    __syncthreads();
    const float wsum = wsum_s;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = threadIdx.x + it * 256;
        if (c >= DV) break;
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < kPre; ++s) {
            const float w = s < splits ? wgt[s] : 0.f;
            if (w != 0.f) acc += w * pre[it][s];
        }
        for (int s = kPre; s < splits; ++s) {
            const float w = wgt[s];
            if (w != 0.f) acc += w * src[s * (DV + 1) + c];
        }
        o[(int64_t)b * o_s0 + (int64_t)h * o_s1 + c] = f32_to_bf16_bits(wsum > 0.f ? acc / wsum : 0.f);
    }
    for (int c = threadIdx.x + 512; c < DV; c += 256) {   // wider value heads than any built today: the plain walk
        float acc = 0.f;
        for (int s = 0; s < splits; ++s) {
            const float w = wgt[s];
            if (w != 0.f) acc += w * src[s * (DV + 1) + c];
        }
        o[(int64_t)b * o_s0 + (int64_t)h * o_s1 + c] = f32_to_bf16_bits(wsum > 0.f ? acc / wsum : 0.f);
    }
}

}  // namespace attn
}  // namespace sglk

using namespace sglk;
using namespace sglk::attn;

extern "C" int sglk_extend_attention(const sglk_extend_attention_args* a, void* stream) {
    SGLK_REQUIRE(a, SGLK_ERR_INVALID, "extend_attention: null args");
    SGLK_REQUIRE(a->B >= 0 && a->HQ > 0 && a->HKV > 0 && a->HQ % a->HKV == 0, SGLK_ERR_INVALID,
                 "extend_attention: bad head counts HQ=%d HKV=%d", a->HQ, a->HKV);
    SGLK_REQUIRE(a->HBUF == a->HKV || a->HBUF == 1, SGLK_ERR_SHAPE, "extend_attention: buffer heads (%d) must equal HKV (%d) or 1", a->HBUF, a->HKV);
    if (a->B == 0 || a->max_len_extend <= 0) return SGLK_OK;
    SGLK_REQUIRE(a->q && a->k_extend && a->v_extend && a->o && a->k_buffer && a->v_buffer && a->req_to_tokens && a->b_req_idx &&
                     a->b_seq_len && a->b_seq_len_extend && a->b_start_loc_extend,
                 SGLK_ERR_INVALID, "extend_attention: null pointer");
    ExtendParams p{};
    p.q = (const unsigned short*)a->q; p.k_ext = (const unsigned short*)a->k_extend; p.v_ext = (const unsigned short*)a->v_extend;
    p.k_buf = (const unsigned short*)a->k_buffer; p.v_buf = (const unsigned short*)a->v_buffer; p.o = (unsigned short*)a->o;
    p.q_s0 = a->q_stride[0]; p.q_s1 = a->q_stride[1]; p.ke_s0 = a->k_extend_stride[0]; p.ke_s1 = a->k_extend_stride[1];
    p.ve_s0 = a->v_extend_stride[0]; p.ve_s1 = a->v_extend_stride[1]; p.kb_s0 = a->k_buffer_stride[0]; p.kb_s1 = a->k_buffer_stride[1];
    p.vb_s0 = a->v_buffer_stride[0]; p.vb_s1 = a->v_buffer_stride[1]; p.o_s0 = a->o_stride[0]; p.o_s1 = a->o_stride[1];
    p.req_to_tokens = a->req_to_tokens; p.rtt_stride = a->req_to_tokens_stride; p.rtt_is64 = a->req_to_tokens_is64;
    p.b_req_idx = a->b_req_idx; p.b_seq_len = a->b_seq_len; p.b_seq_len_extend = a->b_seq_len_extend;
    p.b_start_loc_extend = a->b_start_loc_extend; p.HQ = a->HQ; p.HKV = a->HKV; p.HBUF = a->HBUF;
    p.sm_scale = a->sm_scale; p.logit_cap = a->logit_cap;
    p.d_real = a->D; p.dv_real = a->DV; p.v_aligned = 1;
    // 16-byte row accesses
    const int64_t strides[] = {p.q_s0, p.q_s1, p.ke_s0, p.ke_s1, p.ve_s0, p.ve_s1, p.kb_s0, p.kb_s1, p.vb_s0, p.vb_s1};
    for (int64_t st : strides) SGLK_REQUIRE(st % 8 == 0, SGLK_ERR_SHAPE, "extend_attention: q/k/v strides must be multiples of 8 elements");
    SGLK_REQUIRE(p.o_s0 % 4 == 0 && p.o_s1 % 4 == 0, SGLK_ERR_SHAPE, "extend_attention: o strides must be multiples of 4 elements");
    hipStream_t s = (hipStream_t)stream;
    p.B = a->B; p.n_cu = attn_cus(); p.order = attn_order(2);
    // SGLK_ATTN_NW=4: 4-wave workgroups (twice as many, half the queries each).  Measured at B = 1, 4096 x 32 heads: 0.275 vs
    // 0.231 ms -- the better balance of the causal launch does not pay for two workgroups per CU each staging their own K/V
    // tiles; kept as an A/B knob, never chosen by default.
#define EXT_CASE(DD, DDV)                                                                              \
    if (a->D == DD && a->DV == DDV) {                                                                  \
        constexpr int QT = DD > 128 ? 1 : 2;                                                           \
        const int64_t wgs8 = ceil_div(a->max_len_extend, 8 * QT * 16) * a->B * a->HQ;                  \
        const bool nw4 = knobs().attn_nw == 4; (void)wgs8;                                             \
        const int nw = nw4 ? 4 : 8;                                                                    \
        p.nqblk = (int)ceil_div(a->max_len_extend, nw * QT * 16);                                      \
        p.pair = pair_blocks(p.nqblk, (int64_t)p.nqblk * a->B * a->HQ, p.n_cu, nw4 ? 0 : 1);           \
        const int64_t wgs = (int64_t)(p.pair ? (p.nqblk + 1) / 2 : p.nqblk) * a->B * a->HQ;            \
        SGLK_REQUIRE(wgs < (1ll << 31), SGLK_ERR_SHAPE, "extend_attention: too many workgroups");      \
        const dim3 grid((unsigned)wgs);                                                                \
        if (nw4) hipLaunchKernelGGL((extend_attention_kernel<DD, DDV, QT, 0, 4>), grid, dim3(256), 0, s, p); \
        else hipLaunchKernelGGL((extend_attention_kernel<DD, DDV, QT, 0, 8>), grid, dim3(512), 0, s, p);  \
        SGLK_CHECK_LAUNCH("extend_attention");                                                         \
        return SGLK_OK;                                                                                \
    }
    if ((a->D == 128 || a->D == 192) && a->DV == 128 && knobs().attn_pp != 0 && knobs().attn_nw != 4) {   // two-phase form (SGLK_ATTN_PP=0: the form above)
        p.nqblk = (int)ceil_div(a->max_len_extend, 256);
        p.pair = 0;
        const int64_t wgs = (int64_t)p.nqblk * a->B * a->HQ;
        SGLK_REQUIRE(wgs < (1ll << 31), SGLK_ERR_SHAPE, "extend_attention: too many workgroups");
        if (a->D == 128) {
            constexpr int kPpLds = 3 * kKeys * (128 + 128) * 2;
            SGLK_ENSURE_DYN_LDS((extend_pp_kernel<0, 128>), kPpLds, "extend_attention");
            hipLaunchKernelGGL((extend_pp_kernel<0, 128>), dim3((unsigned)wgs), dim3(512), kPpLds, s, p);
        } else {
            constexpr int kPpLds = 3 * kKeys * (192 + 128) * 2;
            SGLK_ENSURE_DYN_LDS((extend_pp_kernel<0, 192>), kPpLds, "extend_attention");
            hipLaunchKernelGGL((extend_pp_kernel<0, 192>), dim3((unsigned)wgs), dim3(512), kPpLds, s, p);
        }
        SGLK_CHECK_LAUNCH("extend_attention");
        return SGLK_OK;
    }
    EXT_CASE(128, 128)
    EXT_CASE(128, 96)
    EXT_CASE(192, 128)
    EXT_CASE(64, 64)
#undef EXT_CASE
    SGLK_FAIL(SGLK_ERR_SHAPE, "extend_attention: head dims D=%d DV=%d not built (have 128/128, 128/96, 192/128, 64/64)", a->D, a->DV);
}

// flash_attn_varlen_func (/root/reference/test_flash_attn_varlen.py:100-108; oracle flash_attn_varlen_ref :14-46): plain
// variable-length attention without a paged prefix, on the extend kernel.  Head dims that are not multiples of 32 / 16
// (the reference tests 72, 80, 94) run on zero-padded LDS images: q.k is unchanged by zero columns, and the extra value
// columns are never stored.
extern "C" int sglk_flash_attn_varlen(const sglk_flash_attn_varlen_args* a, void* stream) {
    SGLK_REQUIRE(a, SGLK_ERR_INVALID, "flash_attn_varlen: null args");
    SGLK_REQUIRE(a->B >= 0 && a->HQ > 0 && a->HKV > 0 && a->HQ % a->HKV == 0, SGLK_ERR_INVALID,
                 "flash_attn_varlen: bad head counts HQ=%d HKV=%d", a->HQ, a->HKV);
    if (a->B == 0 || a->max_seqlen_q <= 0) return SGLK_OK;
    SGLK_REQUIRE(a->q && a->k && a->v && a->o && a->cu_seqlens_q && a->cu_seqlens_k, SGLK_ERR_INVALID, "flash_attn_varlen: null pointer");
    SGLK_REQUIRE(a->D > 0 && a->D % 8 == 0 && a->D <= 128 && a->DV > 0 && a->DV % 2 == 0 && a->DV <= 128, SGLK_ERR_SHAPE,
                 "flash_attn_varlen: head dims D=%d (multiple of 8, <= 128) DV=%d (even, <= 128) not supported", a->D, a->DV);
    ExtendParams p{};
    p.q = (const unsigned short*)a->q; p.k_ext = (const unsigned short*)a->k; p.v_ext = (const unsigned short*)a->v;
    p.k_buf = p.k_ext; p.v_buf = p.v_ext; p.o = (unsigned short*)a->o;
    p.q_s0 = a->q_stride[0]; p.q_s1 = a->q_stride[1]; p.ke_s0 = a->k_stride[0]; p.ke_s1 = a->k_stride[1];
    p.ve_s0 = a->v_stride[0]; p.ve_s1 = a->v_stride[1]; p.kb_s0 = p.ke_s0; p.kb_s1 = p.ke_s1; p.vb_s0 = p.ve_s0; p.vb_s1 = p.ve_s1;
    p.o_s0 = a->o_stride[0]; p.o_s1 = a->o_stride[1];
    p.HQ = a->HQ; p.HKV = a->HKV; p.HBUF = a->HKV;
    p.sm_scale = a->sm_scale; p.logit_cap = 0.f;
    p.varlen = 1; p.causal = a->causal; p.cu_q = a->cu_seqlens_q; p.cu_k = a->cu_seqlens_k;
    p.d_real = a->D; p.dv_real = a->DV;
    SGLK_REQUIRE(p.q_s0 % 8 == 0 && p.q_s1 % 8 == 0 && p.ke_s0 % 8 == 0 && p.ke_s1 % 8 == 0 && ((uintptr_t)a->q % 16) == 0 &&
                     ((uintptr_t)a->k % 16) == 0, SGLK_ERR_SHAPE, "flash_attn_varlen: q/k rows must be 16-byte aligned");
    SGLK_REQUIRE(p.ve_s0 % 2 == 0 && p.ve_s1 % 2 == 0 && p.o_s0 % 2 == 0 && p.o_s1 % 2 == 0 && ((uintptr_t)a->v % 4) == 0 &&
                     ((uintptr_t)a->o % 4) == 0, SGLK_ERR_SHAPE, "flash_attn_varlen: v/o rows must be 4-byte aligned");
    p.v_aligned = (a->DV % 8 == 0 && p.ve_s0 % 8 == 0 && p.ve_s1 % 8 == 0 && p.o_s0 % 4 == 0 && p.o_s1 % 4 == 0 &&
                   ((uintptr_t)a->v % 16) == 0 && ((uintptr_t)a->o % 8) == 0) ? 1 : 0;
    const int Dp = (a->D + 31) / 32 * 32, DVp = (a->DV + 15) / 16 * 16;
    const bool exact = a->D == Dp && a->DV == DVp && p.v_aligned;
    p.B = a->B; p.n_cu = attn_cus(); p.order = attn_order(2);
    p.nqblk = (int)ceil_div(a->max_seqlen_q, 8 * 2 * 16);
    p.pair = pair_blocks(p.nqblk, (int64_t)p.nqblk * a->B * a->HQ, p.n_cu, a->causal);
    const int64_t wgs = (int64_t)(p.pair ? (p.nqblk + 1) / 2 : p.nqblk) * a->B * a->HQ;
    SGLK_REQUIRE(wgs < (1ll << 31), SGLK_ERR_SHAPE, "flash_attn_varlen: too many workgroups");
    const dim3 block(512);
    hipStream_t s = (hipStream_t)stream;
    if (exact && Dp == 128 && DVp == 128 && knobs().attn_pp != 0) {   // two-phase form
        p.pair = 0;
        const int64_t wgs2 = (int64_t)p.nqblk * a->B * a->HQ;
        constexpr int kPpLds = 6 * kKeys * 128 * 2;
        SGLK_ENSURE_DYN_LDS(extend_pp_kernel<1>, kPpLds, "flash_attn_varlen");
        hipLaunchKernelGGL(extend_pp_kernel<1>, dim3((unsigned)wgs2), block, kPpLds, s, p);
        SGLK_CHECK_LAUNCH("flash_attn_varlen");
        return SGLK_OK;
    }
#define FA_CASE(DD, DDV)                                                                               \
    if (Dp == DD && DVp == DDV) {                                                                      \
        const dim3 grid((unsigned)wgs);                                                                \
        if (exact) hipLaunchKernelGGL((extend_attention_kernel<DD, DDV, 2, 1>), grid, block, 0, s, p); \
        else hipLaunchKernelGGL((extend_attention_kernel<DD, DDV, 2, 2>), grid, block, 0, s, p);       \
        SGLK_CHECK_LAUNCH("flash_attn_varlen");                                                        \
        return SGLK_OK;                                                                                \
    }
    FA_CASE(64, 64)
    FA_CASE(64, 80)
    FA_CASE(64, 96)
    FA_CASE(96, 80)
    FA_CASE(96, 96)
    FA_CASE(128, 96)
    FA_CASE(128, 128)
#undef FA_CASE
    SGLK_FAIL(SGLK_ERR_SHAPE, "flash_attn_varlen: padded head dims D=%d DV=%d not built", Dp, DVp);
}

extern "C" int sglk_decode_attention(const sglk_decode_attention_args* a, void* stream) {
    SGLK_REQUIRE(a, SGLK_ERR_INVALID, "decode_attention: null args");
    SGLK_REQUIRE(a->B >= 0 && a->HQ > 0 && a->HKV > 0 && a->HQ % a->HKV == 0, SGLK_ERR_INVALID,
                 "decode_attention: bad head counts HQ=%d HKV=%d", a->HQ, a->HKV);
    SGLK_REQUIRE(a->HQ / a->HKV <= 64, SGLK_ERR_SHAPE, "decode_attention: at most 64 q heads per kv head");
    SGLK_REQUIRE(a->splits > 0 && a->splits <= 64, SGLK_ERR_INVALID, "decode_attention: attn_logits must have 1 .. 64 splits");
    if (a->B == 0) return SGLK_OK;
    SGLK_REQUIRE(a->q && a->k_buffer && a->v_buffer && a->o && a->key && a->value && a->loc && a->attn_logits && a->req_to_token &&
                     a->b_req_idx && a->b_seq_len, SGLK_ERR_INVALID, "decode_attention: null pointer");
    const int64_t strides[] = {a->q_stride[0], a->q_stride[1], a->k_buffer_stride[0], a->k_buffer_stride[1],
                               a->v_buffer_stride[0], a->v_buffer_stride[1]};
    for (int64_t st : strides) SGLK_REQUIRE(st % 8 == 0, SGLK_ERR_SHAPE, "decode_attention: q/k/v strides must be multiples of 8 elements");
    hipStream_t s = (hipStream_t)stream;
    // V aliases K when it is the same storage with the same strides (MLA: v = k[..., :DV])
    const bool alias = a->v_buffer == a->k_buffer && a->v_buffer_stride[0] == a->k_buffer_stride[0] &&
                       a->v_buffer_stride[1] == a->k_buffer_stride[1] && a->DV <= a->D;
    // The cache write rides in the attention kernel (DecodeParams::fold_write) while the rows every workgroup then has to write --
    // all B requests' -- stay small: one launch and one kernel boundary less.  Measured on one box (profiles/r03_ab_decode.txt):
    // MLA B = 1: 17.5 -> 15.5 us, GQA B = 8 x 8192 keys: 32.4 -> 30.8; neutral at GQA B = 16 x 2048 (8 MB written in all), 3 %
    // SLOWER at MLA B = 40 x 1064 (46 KB per workgroup, 11 MB in all) and 5-8 % slower at GQA B = 64 x 4096 (2048 workgroups x
    // 32 KB): hence the cap on the total.
    // Needs 16-byte rows and the double-buffered kernel form.  SGLK_DEC_FOLD=0: always the separate launch.
    const int64_t fold_bytes = (int64_t)a->B * (a->D + (alias ? 0 : a->DV)) * 2;   // per workgroup; every workgroup writes them
    bool fold = knobs().dec_fold != 0 && fold_bytes <= 48 * 1024 && fold_bytes * a->B * a->HKV * a->splits <= (4ll << 20) &&
                2 * ((size_t)kKeys * a->D * 2 + (alias ? 0 : (size_t)kKeys * a->DV * 2)) <= SGLK_DEC_DOUBLE_LIMIT;
    for (int64_t st : {a->key_stride[0], a->key_stride[1], a->value_stride[0], a->value_stride[1]}) fold = fold && st % 8 == 0;
    for (const void* ptr : {(const void*)a->key, (const void*)a->value, (const void*)a->k_buffer, (const void*)a->v_buffer})
        fold = fold && ((uintptr_t)ptr % 16) == 0;
    if (!fold)
    hipLaunchKernelGGL(kv_cache_write_kernel, dim3((unsigned)a->B, (unsigned)a->HKV), dim3(256), 0, s,
                       (unsigned short*)a->k_buffer, a->k_buffer_stride[0], a->k_buffer_stride[1], (unsigned short*)a->v_buffer,
                       a->v_buffer_stride[0], a->v_buffer_stride[1], (const unsigned short*)a->key, a->key_stride[0],
                       a->key_stride[1], (const unsigned short*)a->value, a->value_stride[0], a->value_stride[1], a->loc,
                       a->loc_is64, a->B, a->HKV, a->D, a->DV);
    SGLK_CHECK_LAUNCH("decode_attention(cache write)");
    DecodeParams p{};
    p.q = (const unsigned short*)a->q; p.k_buf = (const unsigned short*)a->k_buffer; p.v_buf = (const unsigned short*)a->v_buffer;
    p.q_s0 = a->q_stride[0]; p.q_s1 = a->q_stride[1]; p.kb_s0 = a->k_buffer_stride[0]; p.kb_s1 = a->k_buffer_stride[1];
    p.vb_s0 = a->v_buffer_stride[0]; p.vb_s1 = a->v_buffer_stride[1]; p.logits = a->attn_logits;
    p.req_to_token = a->req_to_token; p.rtt_stride = a->req_to_token_stride; p.rtt_is64 = a->req_to_token_is64;
    p.b_req_idx = a->b_req_idx; p.b_seq_len = a->b_seq_len; p.HQ = a->HQ; p.HKV = a->HKV;
    // Splits actually used (<= the caller's scratch depth; the merge reads only these): every (request, kv head, split) is one
    // workgroup and a wide-head (D >= 256) workgroup owns a CU, so more workgroups than CUs run in ROUNDS that each pay the
    // prologue (page lookups, first tile) again.  Use the most splits that still fit one round (SGLK_DEC_SPLITS=n forces n, -1 =
    // all).  Measured (MLA, D = 576): B = 40 x 1064 keys 45.7 -> 32.9 us, B = 128 x 4096 keys 177 -> 134 us (3.4 -> 4.5 TB/s).
    int eff = a->splits;
    {
        const int64_t per_split = (int64_t)a->B * a->HKV;
        const int cus = attn_cus();
        if (knobs().dec_splits > 0) eff = knobs().dec_splits < a->splits ? knobs().dec_splits : a->splits;
        else if (knobs().dec_splits == 0 && a->D >= 256 && per_split * eff > cus) {   // narrow heads: several workgroups share a CU
            eff = (int)(cus / per_split);
            if (eff < 1) eff = 1;
        } else if (knobs().dec_splits == 0 && a->D < 256 && per_split * eff > (int64_t)8 * cus) {
            // narrow heads with many (request, kv head) pairs: beyond ~8 workgroups per CU more splits only add prologues (page
            // lookups, first tile) -- 40 requests x 22 MHA heads x 8 splits of a 33-key sequence were 7040 workgroups and 52 us
            eff = (int)((int64_t)8 * cus / per_split);
            if (eff < 1) eff = 1;
        }
    }
    p.splits = eff;
    p.logit_splits = a->splits;
    p.sm_scale = a->sm_scale; p.logit_cap = a->logit_cap;
    p.abl = knobs().rescale_ablate;   // read by SGLK_DEV_ABLATE builds only
    // Each cache row is read once, by one workgroup: the non-temporal policy keeps it from displacing anything in L2 / the Infinity
    // Cache.  Measured on one box against the default policy (hipGraph replays, profiles/r03_ab_decode.txt): MLA B = 128 x 4096 keys
    // 0.128 -> 0.120 ms, GQA B = 64 x 4096 0.110 -> 0.097 ms, and still 0.0278 -> 0.0253 ms at B = 40 x 1064, whose 49 MB would fit the
    // Infinity Cache between replays.  SGLK_DEC_NT=0 turns it off.
    p.nt = knobs().dec_nt != 0 ? 1 : 0;
    p.key = (const unsigned short*)a->key; p.value = (const unsigned short*)a->value; p.loc = a->loc; p.loc_is64 = a->loc_is64;
    p.key_s0 = a->key_stride[0]; p.key_s1 = a->key_stride[1]; p.val_s0 = a->value_stride[0]; p.val_s1 = a->value_stride[1];
    p.B = a->B; p.fold_write = fold ? 1 : 0;
    p.o = (unsigned short*)a->o; p.o_s0 = a->o_stride[0]; p.o_s1 = a->o_stride[1];
    p.direct = (eff == 1 && a->o_stride[0] % 4 == 0 && a->o_stride[1] % 4 == 0 && ((uintptr_t)a->o % 8) == 0) ? 1 : 0;
    const dim3 grid((unsigned)a->B, (unsigned)a->HKV, (unsigned)eff);
    const int group = a->HQ / a->HKV;
#define DEC_LAUNCH(DD, DDV, AL, ND)                                                                                \
    {                                                                                                              \
        constexpr size_t kb = (size_t)kKeys * DD * 2, vb = (AL) ? 0 : (size_t)kKeys * DDV * 2;                     \
        constexpr size_t lds = (2 * (kb + vb) <= SGLK_DEC_DOUBLE_LIMIT) ? 2 * (kb + vb) + 4 * kKeys * 4 : (kb + vb);  /* + id tables */ \
        constexpr int kh = DD >= 256 ? SGLK_DEC_KH_WIDE : 1;   /* wide heads: the tile's keys are split over two wave groups */ \
        SGLK_ENSURE_DYN_LDS((decode_attention_kernel<DD, DDV, AL, ND, kh>), lds, "decode_attention");           \
        hipLaunchKernelGGL((decode_attention_kernel<DD, DDV, AL, ND, kh>), grid, dim3(256 * kh), lds, s, p);       \
    }
#define DEC_CASE(DD, DDV)                                                                                          \
    if (a->D == DD && a->DV == DDV) {                                                                              \
        if (alias) {                                                                                               \
            if (group <= 16) DEC_LAUNCH(DD, DDV, true, 4)                                                          \
            else if (group <= 32) DEC_LAUNCH(DD, DDV, true, 2)                                                     \
            else DEC_LAUNCH(DD, DDV, true, 1)                                                                      \
        } else {                                                                                                   \
            if (group <= 16) DEC_LAUNCH(DD, DDV, false, 4)                                                         \
            else if (group <= 32) DEC_LAUNCH(DD, DDV, false, 2)                                                    \
            else DEC_LAUNCH(DD, DDV, false, 1)                                                                     \
        }                                                                                                          \
        SGLK_CHECK_LAUNCH("decode_attention");                                                                     \
        if (p.direct) return SGLK_OK;                                                                              \
        hipLaunchKernelGGL(decode_merge_kernel, dim3((unsigned)a->B, (unsigned)a->HQ), dim3(256), 0, s, a->attn_logits,    \
                           (unsigned short*)a->o, a->o_stride[0], a->o_stride[1], a->HQ, eff, a->splits, a->DV);   \
        SGLK_CHECK_LAUNCH("decode_attention(merge)");                                                              \
        return SGLK_OK;                                                                                            \
    }
    DEC_CASE(576, 512)
    DEC_CASE(128, 128)
    DEC_CASE(192, 128)
    DEC_CASE(64, 64)
#undef DEC_CASE
#undef DEC_LAUNCH
    SGLK_FAIL(SGLK_ERR_SHAPE, "decode_attention: head dims D=%d DV=%d not built (have 576/512, 128/128, 192/128, 64/64)", a->D, a->DV);
}
