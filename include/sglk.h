/*
 * sglk — C-ABI of the MI355X (gfx950) sgl_kernel hot path.
 *
 * This is the drop-in boundary.  The reference harness reaches its kernels through
 * `torch.ops.sgl_kernel.<op>` (e.g. /root/reference/bench_moe.py:5-6); the library behind those ops is the
 * absent third-party CPU build of sgl_kernel.  Every entry point below is what that operator's binding
 * would call for this path: plain pointers, sizes, strides and a HIP stream — no torch types.
 * The Python package `sgl_kernel` (sgl-cpu-tests_amd/sgl_kernel) registers the reference's operator
 * signatures with torch.library and forwards to these symbols through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name says host; the caller owns all memory, including the
 *     workspace (query the size with the matching *_workspace_bytes function; 256-byte aligned);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it, nothing
 *     synchronises, allocates or frees (hipGraph-capture safe).  The library keeps NO streams, events or device memory
 *     of its own: the only process state is the thread-local error string, the environment knobs read once
 *     (sglk_reload_env) and per-device caches of two device properties (CU count, "dynamic LDS size raised" bits);
 *   - calls act on the CURRENT device: make the device that owns `stream` and the buffers current before calling;
 *   - return value: 0 on success, negative SGLK_ERR_* otherwise; sglk_last_error() gives a thread-local
 *     message.  Shape/divisibility violations are rejected before anything is launched;
 *   - bf16 = uint16_t bit pattern, fp8 = OCP e4m3fn byte, strides in ELEMENTS.
 */
#ifndef SGLK_H_
#define SGLK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGLK_VERSION 200 /* 0.2.0 */

enum {
    SGLK_OK = 0,
    SGLK_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, unsupported flag) */
    SGLK_ERR_SHAPE = -2,       /* shape / divisibility contract violated */
    SGLK_ERR_WORKSPACE = -3,   /* workspace too small */
    SGLK_ERR_LAUNCH = -4       /* HIP reported a launch error */
};

/* weight element types accepted by the pack / GEMM entry points */
enum {
    SGLK_W_BF16 = 0,
    SGLK_W_FP8_E4M3 = 1,
    SGLK_W_INT8 = 2,
    SGLK_W_MXFP4 = 3   /* E2M1 pairs in bytes + E8M0 scales per 32 (sglk_mxfp4_scaled_mm only) */
};

/* output element types of the dense GEMM entry points */
enum {
    SGLK_OUT_BF16 = 0,
    SGLK_OUT_F16 = 1,
    SGLK_OUT_F32 = 2
};

int sglk_version(void);
const char* sglk_last_error(void);
/* number of compute units / name of device `dev` (host-side query used by the bench for roofline peaks) */
int sglk_device_cu_count(int dev);
/* Developer / test knobs (SGLK_* environment variables, listed in sgl-cpu-tests_amd/csrc/knobs.h) are read once, at
 * first use.  This re-reads them; not thread-safe against concurrent calls into the library (tests call it between
 * launches to switch tilings inside one process). */
void sglk_reload_env(void);
/* Convenience for callers that want sglk_fused_experts' optional second stream: creates a non-blocking stream and two
 * timing-disabled events on the current device / destroys them.  The caller owns them; the library keeps no reference. */
int sglk_aux_create(void** stream /* host out */, void** event0 /* host out */, void** event1 /* host out */);
void sglk_aux_destroy(void* stream, void* event0, void* event1);

/* ---------------------------------------------------------------------------------------------------------
 * convert_weight_packed        replaces torch.ops.sgl_kernel.convert_weight_packed
 *                              (/root/reference/bench_moe.py:26-27,43-44; test_moe_fp8_ext.py:114-115)
 * Re-tiles `batch` row-major matrices [rows][cols] into the MFMA-operand tile order the GEMM kernels stream
 * (DESIGN.md §Packed weight layout).  Output has the same byte size, dtype and shape as the input, so it
 * stays an ordinary clonable tensor exactly like the reference's packed weights (bench_moe.py:53-58).
 * Requirements: fp8/int8 rows % 16 == 0 and cols % 64 == 0; bf16 rows % 32 == 0 and cols % 8 == 0 (bf16 keeps the
 * reference's VNNI-2 order, the one layout it pins with a live known-answer test, test_gemm.py:36-46).
 * --------------------------------------------------------------------------------------------------------- */
int sglk_pack_weight(const void* src, void* dst, int64_t batch, int64_t rows, int64_t cols, int wtype,
                     void* stream);
/* inverse of sglk_pack_weight (used by tests: pack -> unpack must be the identity) */
int sglk_unpack_weight(const void* src, void* dst, int64_t batch, int64_t rows, int64_t cols, int wtype,
                       void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * fused_experts               replaces torch.ops.sgl_kernel.fused_experts_cpu
 *     14-arg form /root/reference/bench_moe.py:113-130, test_moe_fp8_ext.py:118,
 *     test_moe_offloading_cpu.py:103-117; 13-arg form /root/reference/test_moe.py:79-92.
 *
 *   out[m] = sum_j topk_weights[m,j] * ( silu(x_m W1g[e]^T) * (x_m W1u[e]^T) ) W2[e]^T ,  e = topk_ids[m,j]
 *   slots with topk_ids outside [0,E) (the reference's -1 padding) contribute nothing.
 *
 * hidden  [M][K] bf16, row stride hidden_stride        out [M][K] bf16 (may alias hidden: `inplace`)
 * w1      [E][2N][K]  rows [0,N) gate, [N,2N) up       w2  [E][K][N]
 * wtype   SGLK_W_FP8_E4M3: W8A16, w*_scale f32 [E][ceil(rows/block_n)][ceil(cols/block_k)], block_k == 128
 *         SGLK_W_BF16    : scales ignored
 *         SGLK_W_INT8    : W8A8 dynamic per-token activation quant, w1_scale [E][2N], w2_scale [E][K]
 * packed  bit 0: w1 is in sglk_pack_weight order, bit 1: w2 is (reference `is_vnni=True` = 3 when both shapes can be
 *         packed); 0: plain row-major.  The tuned fp8 kernels need both bits; anything else runs the generic engine
 * topk_weights [M][topk] f32, topk_ids [M][topk] i32
 * --------------------------------------------------------------------------------------------------------- */
typedef struct {
    const void* hidden;
    int64_t hidden_stride;
    void* out;
    int64_t out_stride;
    const void* w1;
    const void* w2;
    const float* w1_scale;
    const float* w2_scale;
    const float* topk_weights;
    const int32_t* topk_ids;
    int32_t M, N, K, E, topk;
    int32_t wtype;
    int32_t packed;
    int32_t block_n, block_k;
    void* workspace;
    size_t workspace_bytes;
    void* stage_timer; /* optional sglk_stage_timer (NULL = off): records HIP events around every stage */
    /* Optional second stream (hipStream_t) + two events (hipEvent_t, timing disabled) of the caller: when all three are
     * given, the short tail tiles of a mid-size batch run on `aux_stream`, forked after the routing sort and joined before
     * the combine (both inside this call, hipGraph-capture safe).  NULL: everything runs on `stream`. */
    void* aux_stream;
    void* aux_events[2];
    int32_t flags;          /* SGLK_MOE_* bits */
    int32_t* path_taken;    /* optional host out: which kernels this call used (SGLK_PATH_*), for measurement reports */
} sglk_fused_experts_args;

/* flags */
enum {
    /* opt-in "a8" mode of the fp8 path (NOT the reference's W8A16 numerics; never a default): activations are quantised
     * per token x 128-wide block to e4m3 with a power-of-two scale and both GEMMs run on the block-scaled fp8 matrix
     * cores (v_mfma_scale_f32_32x32x64_f8f6f4).  Needs packed weights, block [128k,128], K % 256 == 0, N % 128 == 0.
     * Stated tolerance: DESIGN.md "a8 mode". */
    SGLK_MOE_FP8_ACT = 1,
    /* the weights are ROW-MAJOR (packed == 0) and the workspace (sized with _workspace_bytes_ex and this flag) has room for a
     * re-tiled copy: the call packs both weights into the workspace and runs the packed kernels.  One pass over the weight bytes
     * per call -- worth it from ~64 rows per expert on, where the tuned kernels are 3-4x the generic engine; a caller that
     * holds its weights packed (the reference's is_vnni=True, bench_moe.py:26-27) never needs it. */
    SGLK_MOE_PACK_WEIGHTS = 2
};
/* path_taken: tile height of the grouped GEMMs (32 / 96 / 128 / 256; 64 = generic engine) | flag bits */
enum {
    SGLK_PATH_TILE_MASK = 0x3ff,
    SGLK_PATH_FP8_ACT = 0x1000,       /* a8 mode (moe_gemm_fp8w_s128.hip, one e4m3 term) */
    SGLK_PATH_TAILS_SPLIT = 0x2000,   /* tail tiles on the mid kernel */
    SGLK_PATH_TAILS_AUX = 0x4000,     /* ... on the caller's aux stream */
    SGLK_PATH_PERSIST_G1 = 0x8000,    /* GEMM-1 launched persistent (one workgroup per CU, tile loop + tickets) */
    SGLK_PATH_PERSIST_G2 = 0x10000,   /* GEMM-2 launched persistent */
    SGLK_PATH_ROUTE_ALIGN = 0x20000,  /* sglk_moe_block: router + align ran as one launch */
    SGLK_PATH_SHARED_FOLDED = 0x40000,/* sglk_moe_block: routed combine folded into the shared expert's last launch */
    SGLK_PATH_SPLIT = 0x80000,        /* W8A16 with the activations as two exact e4m3 terms on the scaled fp8 MFMA */
    SGLK_PATH_INLINE_ALIGN = 0x100000 /* at most 32 slots: no moe_align launch, the GEMM workgroups sort the ids themselves */
};

size_t sglk_fused_experts_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t E, int32_t topk, int32_t wtype);
/* the same with the SGLK_MOE_* flags of the call (the a8 mode keeps quantised copies of the activations) */
size_t sglk_fused_experts_workspace_bytes_ex(int32_t M, int32_t N, int32_t K, int32_t E, int32_t topk, int32_t wtype,
                                             int32_t flags);
int sglk_fused_experts(const sglk_fused_experts_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * moe_block                   router -> routed experts (-> shared expert) as ONE call
 * The reference harness makes these calls one after the other (/root/reference/test_moe.py:57-92:
 * grouped_topk_cpu then fused_experts_cpu; /root/reference/test_shared_experts.py:34-40,68: shared_expert_cpu with the
 * routed output as `fused_experts_out`); this entry point computes the same thing with fewer launches:
 *   topk_weights, topk_ids = grouped_topk(gating, topk, renormalize, num_expert_group, topk_group [, correction_bias])
 *   routed = fused_experts(hidden, w1, w2, topk_weights, topk_ids)
 *   out    = shared_N > 0 ? shared_mlp(hidden; shared_w1, shared_w2) + routed * routed_scaling_factor : routed
 * - decode batches (M <= 16, E <= 256): router and the routing sort are ONE launch; ids / weights are bit-identical to
 *   sglk_grouped_topk's and are returned in experts.topk_weights / experts.topk_ids (OUTPUT buffers [M][topk] here);
 * - shared expert, fp8 packed weights, decode sizes: its last launch sums the routed slots itself (no combine launch, no
 *   [M][K] round trip).  `routed` is then never rounded to bf16, so the result differs from the three separate calls by
 *   that one rounding (it is the more accurate one); other shapes run the separate kernels on a scratch buffer.
 * experts.out is the final output; experts.workspace must hold sglk_moe_block_workspace_bytes().
 * --------------------------------------------------------------------------------------------------------- */
typedef struct {
    sglk_fused_experts_args experts;
    const void* gating;           /* [M][E] router logits */
    int64_t gating_stride;
    int32_t gating_type;          /* 0 bf16, 1 f16, 2 f32 */
    const void* correction_bias;  /* [E] same type as gating, or NULL (softmax scoring) */
    int32_t renormalize, num_expert_group, topk_group;
    int32_t shared_N;             /* shared expert's intermediate size; 0 = no shared expert */
    const void* shared_w1;        /* [2*shared_N][K] */
    const void* shared_w2;        /* [K][shared_N] */
    const float* shared_w1_scale;
    const float* shared_w2_scale;
    int32_t shared_packed;
    float routed_scaling_factor;
} sglk_moe_block_args;
size_t sglk_moe_block_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t E, int32_t topk, int32_t wtype, int32_t flags,
                                      int32_t shared_N);
int sglk_moe_block(const sglk_moe_block_args* args, void* stream);

/* First stage of the opt-in a8 mode (SGLK_MOE_FP8_ACT), exported so that it can be checked bit for bit on its own:
 * x [rows][cols] bf16 -> q [rows][cols] e4m3 + scale [rows][scale_stride] E8M0 bytes, one per 128-wide block
 * (cols % 128 == 0): power-of-two scale 2^(byte-127) = the smallest with amax / scale <= 448, round to nearest even.
 * q is stored in the k order of the packed weight tile (inside every 64 group position 32h+q holds
 * k = 16h + {q | 32+q-8 | 8+q-16 | 40+q-24}), the order sglk_fused_experts' a8 kernels read.  No reference counterpart
 * (the reference's fp8 op keeps bf16 activations, /root/reference/bench_moe.py:113-130). */
int sglk_quant_fp8_block128(const void* x, int64_t x_stride, void* q, int64_t q_stride, void* scale, int64_t scale_stride,
                            int64_t rows, int32_t cols, void* stream);

/* First stage of the W8A16 path on the scaled fp8 matrix cores (moe_gemm_fp8w_s128.hip; fp8_split.h), exported so that the exactness of
 * the split can be checked on its own: x [rows][cols] bf16 -> q [rows][2*cols] bytes + scale [rows][scale_stride] E8M0 bytes.
 * Per 128-wide block: s = 2^(byte-127), the smallest power of two with amax / s <= 448 (byte >= 5); per element
 * hi = e4m3(x / s), lo = e4m3((x - hi * s) / (s / 16)): x == hi * s + lo * s / 16 EXACTLY for every element within 2^13 of
 * its block's amax.  q holds, per 64-wide k group, [hi 64 bytes | lo 64 bytes] in the packed-tile k order (see
 * sglk_quant_fp8_block128).  No reference counterpart (internal stage of fused_experts_cpu, /root/reference/bench_moe.py:113-130). */
int sglk_split_fp8_block128(const void* x, int64_t x_stride, void* q, int64_t q_stride, void* scale, int64_t scale_stride,
                            int64_t rows, int32_t cols, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * shared_expert               replaces torch.ops.sgl_kernel.shared_expert_cpu
 *     14-arg /root/reference/test_moe_fp8.py:87-88, test_moe_fp8_ext.py:60-61; 12-arg test_shared_experts.py:68,78
 *   out = ( silu(x W1g^T) * (x W1u^T) ) W2^T + fused_out * routed_scaling_factor
 * Same operand conventions as fused_experts with E = 1 (w1 [2N][K], w2 [K][N]); fused_out [M][K] bf16.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct {
    const void* hidden;
    int64_t hidden_stride;
    void* out;
    int64_t out_stride;
    const void* w1;
    const void* w2;
    const float* w1_scale;
    const float* w2_scale;
    const void* fused_out;
    int64_t fused_out_stride;
    float routed_scaling_factor;
    int32_t M, N, K;
    int32_t wtype;
    int32_t packed;
    int32_t block_n, block_k;
    void* workspace;
    size_t workspace_bytes;
} sglk_shared_expert_args;

size_t sglk_shared_expert_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t wtype);
/* ... sized so that row-major weights (packed == 0) at prefill sizes are re-tiled into the workspace and run on the packed paths */
size_t sglk_shared_expert_workspace_bytes_ex(int32_t M, int32_t N, int32_t K, int32_t wtype, int32_t packed);
int sglk_shared_expert(const sglk_shared_expert_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Dense weight GEMMs  out[M][N] = x[M][K] . w[N][K]^T (+ bias)
 *   replaces weight_packed_linear (/root/reference/test_gemm.py:22-25)            wtype BF16, x bf16
 *            fp8_scaled_mm_cpu    (/root/reference/test_gemm_fp8.py:54-62)        wtype FP8, block scales, x bf16
 *            int8_scaled_mm_cpu   (/root/reference/test_gemm_int8.py:67)          wtype INT8, x int8 + x_scale[M]
 *            int8_scaled_mm_with_quant (test_gemm_int8.py:72)                     wtype INT8, x bf16 (quantised here)
 * x_is_int8 != 0: x is int8 [M][K] with per-row scales x_scale; else bf16.  w_scale: fp8 [ceil(N/bn)][K/128],
 * int8 [N].  bias [N] f32 or NULL.  out_type SGLK_OUT_*.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct {
    const void* x;
    int64_t x_stride;
    int32_t x_is_int8;
    const float* x_scale;
    const void* w;
    const float* w_scale;
    const float* bias;
    void* out;
    int64_t out_stride;
    int32_t out_type;
    int32_t M, N, K;
    int32_t wtype;
    int32_t packed;
    int32_t block_n, block_k;
    void* workspace;
    size_t workspace_bytes;
} sglk_scaled_mm_args;

size_t sglk_scaled_mm_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t wtype, int32_t x_is_int8);
/* ... plus room for a re-tiled copy of a ROW-MAJOR weight (packed == 0) when M >= 192 and the shape can be packed: the call
 * then packs into the workspace and runs the tuned kernels (3-4x the generic engine at prefill sizes).  With only the bytes of
 * the function above a row-major weight runs on the generic engine at every M. */
size_t sglk_scaled_mm_workspace_bytes_ex(int32_t M, int32_t N, int32_t K, int32_t wtype, int32_t x_is_int8, int32_t packed);
int sglk_scaled_mm(const sglk_scaled_mm_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Direct all-reduce over xGMI peer memory     replaces the ROLE of sgl_kernel.common_ops.shm_allreduce
 *     (/root/reference/test_allreduce.py:86-105: a shared-memory all-reduce between the CPU ranks of one host; bench message
 *      1024 x 5120 bf16 = 10 MiB, run_allreduce_cpu.sh:8-12).  One process per GPU; every rank allocates a staging region of
 *      4 * capacity bytes and 256 bytes of flag words with sglk_comm_alloc, exports both with sglk_ipc_export, exchanges the
 *      64-byte handles out of band (torch.distributed object all-gather in sgl_kernel/collectives.py) and maps its peers' with
 *      sglk_ipc_open.  sglk_allreduce_sum_bf16 then sums `n_elems` bf16 values of all ranks (fp32, ascending rank order, one
 *      rounding: identical bits on every rank) from `in` to `out` (may alias): one-shot for small messages, two-shot
 *      (reduce-scatter + all-gather of slices) for large ones -- direct reads use all 7 links of a GPU at once, a ring is bound
 *      by one.  `epoch` must increase by one per call (the same value on every rank); *status_dev becomes non-zero when a
 *      peer did not arrive within ~2 s (the kernels give up instead of hanging).  algo: 0 by size, 1 one-shot, 2 two-shot.
 * --------------------------------------------------------------------------------------------------------- */
int sglk_comm_alloc(size_t bytes, int32_t finegrained, void** dev_ptr /* host out */);
void sglk_comm_free(void* dev_ptr);
int sglk_ipc_export(const void* dev_ptr, void* handle64 /* host, 64 bytes out */);
int sglk_ipc_open(const void* handle64 /* host */, void** dev_ptr /* host out */);
int sglk_ipc_close(void* dev_ptr);
int sglk_allreduce_sum_bf16(void* const* peer_data /* host [world] */, void* const* peer_flags /* host [world] */, int32_t rank,
                            int32_t world, int64_t capacity_bytes, const void* in, void* out, int64_t n_elems, uint32_t epoch,
                            int32_t algo, int32_t* status_dev, void* stream);

/* Expert-parallel dispatch glue (sgl_kernel/expert_parallel.py; no reference counterpart -- the reference pins only the
 * local contract EP needs, topk_ids == -1 for non-resident experts: /root/reference/test_moe_offloading_cpu.py:12-15,62-68).
 * Rank d owns experts [d*E/G, (d+1)*E/G).
 * sglk_ep_plan: counts[d] = tokens with at least one slot on rank d; pos[m][d] = index of token m among them in ascending
 *   token order (-1: not sent); seg_start[G+1] = first payload row of every destination: exclusive sums of counts
 *   (capacity == 0), or d * capacity (capacity > 0: fixed-size segments, no host read of the counts needed; *overflow gets
 *   bit d when counts[d] > capacity, the surplus tokens are dropped).
 * sglk_ep_pack: payload row (seg_start[d] + pos[m][d]) = [K bf16 of token m | topk ids in d's numbering, -1 elsewhere |
 *   topk routing weights f32], row_bytes >= 2K + 8 topk and a multiple of 16; capacity > 0: ids of unused rows = -1. */
int sglk_ep_plan(const int32_t* topk_ids, int32_t M, int32_t topk, int32_t E, int32_t G, int32_t capacity, int32_t* counts,
                 int32_t* seg_start, int32_t* pos, int32_t* overflow, void* stream);
int sglk_ep_pack(const void* hidden, int64_t hidden_stride, const int32_t* topk_ids, const float* topk_weights,
                 const int32_t* pos, const int32_t* seg_start, const int32_t* counts, void* payload, int64_t row_bytes, int32_t M,
                 int32_t K, int32_t topk, int32_t E, int32_t G, int32_t capacity, void* stream);
/* Expert-parallel combine (no reference counterpart: the exchange around fused_experts):
 * out[m][:] = sum over d = 0..G-1 ascending of rows[table[m*G + d]][:] for table entries >= 0; bf16 rows, fp32 sum, one
 * bf16 rounding.  K % 8 == 0, strides in elements (multiples of 8). */
int sglk_ep_reduce_rows(const void* rows, int64_t rows_stride, const int32_t* table, int32_t G, void* out,
                        int64_t out_stride, int32_t M, int32_t K, void* stream);

/* mxfp4_scaled_mm_cpu (/root/reference/test_mxfp4.py:148,170,201): out[M][N] bf16 = x[M][K] bf16 . dequant(wq, scales)^T
 * (+ bias f32 [N]).  wq uint8 [N][K/2], element 2i of a row in the low nibble of byte i (E2M1: sign, 2-bit exponent,
 * 1-bit mantissa -> 0, 0.5, 1, 1.5, 2, 3, 4, 6); scales uint8 E8M0 (2^(s-127)) per 32 consecutive k: [N][K/32] row-major,
 * or with scale_packed != 0 the order convert_scale_packed returns ([N/32][K/32][32], test_mxfp4.py:186).  The weights
 * are never dequantised to memory and the activations are never quantised (W4A16, like the reference):
 *   M >= 64, N % 128 == 0, K % 256 == 0: v_mfma_scale_f32_32x32x64_f8f6f4 with the E2M1 weights and their E8M0 block scales
 *     as stored; the bf16 activations enter as two e4m3 terms (exact for elements within 2^13 of their 128-block's maximum);
 *   otherwise: exact expansion to bf16 in registers, bf16 matrix cores.
 * fp32 accumulation, one bf16 rounding.  K % 32 == 0; scale_packed needs N % 32 == 0.  Workspace:
 * sglk_mxfp4_workspace_bytes(M, N, K) (with only sglk_scaled_mm_workspace_bytes(M, N, K, SGLK_W_BF16, 0) bytes the second
 * form runs for every shape). */
size_t sglk_mxfp4_workspace_bytes(int32_t M, int32_t N, int32_t K);
int sglk_mxfp4_scaled_mm(const void* x, int64_t x_stride, const void* wq, const void* scales, int32_t scale_packed,
                         const float* bias, void* out, int64_t out_stride, int32_t M, int32_t N, int32_t K,
                         void* workspace, size_t workspace_bytes, void* stream);

/* per_token_quant_int8_cpu (/root/reference/test_gemm_int8.py:66): q = rint(x * 127/amax), scale = amax/127,
 * amax = max(|row|, 1e-10).  x bf16 [rows][cols]. */
int sglk_per_token_quant_int8(const void* x, int64_t x_stride, void* q, int64_t q_stride, float* scale,
                              int64_t rows, int32_t cols, void* stream);

/* the same with an explicit amax floor (the MLA projection's oracle uses 1e-7, /root/reference/test_absorb.py:33-40) */
int sglk_per_token_quant_int8_floor(const void* x, int64_t x_stride, void* q, int64_t q_stride, float* scale,
                                    int64_t rows, int32_t cols, float floor, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * qkv_proj_with_rope(hidden_states, q_a_proj_weight, q_b_proj_weight, kv_a_proj_weight, w_kc, q_a_layernorm_weight,
 *                    kv_a_layernorm_weight, positions, cos_sin_cache, eps, use_int8_w8a8, use_fp8_w8a16, q_a_proj_scale,
 *                    q_b_proj_scale, kv_a_proj_scale, is_vnni, block_size)            /root/reference/test_absorb.py:133-147,184-186
 * in one call (oracles native_torch test_absorb.py:65-87, native_torch_int8 :89-131): the three projections (bf16, fp8 W8A16 with
 * block scales, or int8 W8A8 with per-token activation quantisation, floor 1e-7), the two RMSNorms, the per-head product with
 * w_kc [H][kv_lora][nope], GPT-J rotary embedding of the rope parts.  Every intermediate is rounded to bf16 where the oracle
 * rounds.  Outputs: q_input [B][H][kv_lora + rope], k_input [B][1][kv_lora + rope], v_input [B][1][kv_lora] (bf16; strides in
 * elements).  packed_*: the weight is in sglk_pack_weight order.  Intermediates live in the caller's workspace.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct {
    const void* hidden;          /* [B][hidden_size] bf16 */
    int64_t hidden_stride;
    int32_t B, hidden_size;
    const void *q_a_w, *q_b_w, *kv_a_w;              /* [q_lora][hidden], [H*(nope+rope)][q_lora], [kv_lora+rope][hidden] */
    const float *q_a_scale, *q_b_scale, *kv_a_scale; /* fp8: block scales; int8: per output row; bf16: NULL */
    int32_t wtype;                                   /* SGLK_W_BF16 / SGLK_W_FP8_E4M3 / SGLK_W_INT8 */
    int32_t packed_q_a, packed_q_b, packed_kv_a;
    int32_t block_n, block_k;                        /* fp8 */
    const void* w_kc;                                /* [H][kv_lora][nope] bf16 */
    int32_t w_kc_packed;
    const void *q_a_ln, *kv_a_ln;                    /* RMSNorm weights bf16 [q_lora], [kv_lora] */
    float eps;
    const void* positions;                           /* [B] int32 / int64 */
    int32_t positions_is64;
    const void* cos_sin_cache;                       /* [max_pos][rope] bf16 */
    int64_t cache_stride;
    int32_t H, q_lora, kv_lora, nope, rope;
    void* q_input;
    int64_t q_stride_b, q_stride_h;
    void* k_input;
    int64_t k_stride_b;
    void* v_input;
    int64_t v_stride_b;
    void* workspace;
    size_t workspace_bytes;
} sglk_qkv_proj_args;

size_t sglk_qkv_proj_workspace_bytes(int32_t B, int32_t hidden_size, int32_t H, int32_t q_lora, int32_t kv_lora, int32_t nope,
                                     int32_t rope, int32_t wtype);
int sglk_qkv_proj_with_rope(const sglk_qkv_proj_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Pieces of qkv_proj_with_rope (/root/reference/test_absorb.py:133-147; oracle native_torch :65-87) that are neither a
 * GEMM nor an RMSNorm; the Python layer composes the operator from these, sglk_scaled_mm and sglk_rmsnorm.
 *   bmm_heads  out[b][h][oc] = sum_ic x[b][h][ic] * w[h][oc][ic]   (bf16, fp32 accumulation; w [H][OC][IC] row-major or
 *              packed per head in convert_weight_packed's bf16 order)         torch.bmm(q_nope^T, w_kc), test_absorb.py:73
 *   rope_gptj  out = x * cos + rotate_gptj(x) * sin on the rotary slices of q [B][H][d] and k [B][d]; cache row =
 *              [cos(d/2) | sin(d/2)] indexed by positions                      rotary_emb, test_absorb.py:27-31,49-63
 * strides in elements.
 * --------------------------------------------------------------------------------------------------------- */
int sglk_bmm_heads(const void* x, int64_t x_stride_b, int64_t x_stride_h, const void* w, int32_t packed, void* out,
                   int64_t out_stride_b, int64_t out_stride_h, int32_t B, int32_t H, int32_t OC, int32_t IC, void* stream);
int sglk_rope_gptj(const void* q_pe, int64_t q_stride_b, int64_t q_stride_h, const void* k_pe, int64_t k_stride_b,
                   const void* positions, int32_t positions_is64, const void* cos_sin_cache, int64_t cache_stride,
                   void* q_out, int64_t q_out_stride_b, int64_t q_out_stride_h, void* k_out, int64_t k_out_stride_b,
                   int32_t B, int32_t H, int32_t rotary_dim, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Row kernels (bf16 or fp16 I/O selected by is_f16; strides in elements)
 *   silu_and_mul        out[r][c] = silu(x[r][c]) * x[r][d+c]      /root/reference/test_activation.py:14-28,
 *                                                                  bench_silu_and_mul.py:31
 *   rmsnorm             out = rmsnorm(x) * weight                   /root/reference/test_norm.py:43-47
 *   fused_add_rmsnorm   residual <- x + residual; x <- rmsnorm(residual) * weight (both in place)   test_norm.py:52-59
 * --------------------------------------------------------------------------------------------------------- */
int sglk_silu_and_mul(const void* x, int64_t x_stride, void* out, int64_t out_stride, int64_t rows, int32_t d,
                      int32_t is_f16, void* stream);
int sglk_rmsnorm(void* out, int64_t out_stride, const void* x, int64_t x_stride, const void* weight, int64_t rows,
                 int32_t hidden, float eps, int32_t is_f16, void* stream);
int sglk_fused_add_rmsnorm(void* x, int64_t x_stride, void* residual, int64_t res_stride, const void* weight,
                           int64_t rows, int32_t hidden, float eps, int32_t is_f16, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * grouped_topk / biased_grouped_topk   /root/reference/test_grouped_topk.py:61-69, test_moe.py:61-70,
 *                                      test_biased_grouped_topk.py:71-80
 * gating [M][E] (gating_type 0 bf16, 1 f16, 2 f32); bias [E] of the same type or NULL (NULL = softmax scoring,
 * else sigmoid + bias with top-2-sum group scores).  Writes topk_weights [M][topk] f32 and topk_ids [M][topk] i32
 * in descending order of the selection score; ties -> lower expert index.
 * --------------------------------------------------------------------------------------------------------- */
int sglk_grouped_topk(const void* gating, int64_t gating_stride, int32_t gating_type, const void* bias,
                      float* topk_weights, int32_t* topk_ids, int32_t M, int32_t E, int32_t topk, int32_t renormalize,
                      int32_t num_expert_group, int32_t topk_group, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * extend_attention            replaces torch.ops.sgl_kernel.extend_attention_cpu
 *                             (/root/reference/test_extend.py:168-182, bench_extend.py:70-102)
 * Varlen prefill with a paged prefix: for sequence b the queries are its extend tokens
 * q[b_start_loc_extend[b] + i], keys/values are k_buffer[req_to_tokens[b_req_idx[b]][0 .. prefix)] (all visible,
 * prefix = b_seq_len[b] - b_seq_len_extend[b]) followed by the extend tokens k_extend/v_extend (causal).
 * All tensors bf16 [tokens][heads][dim]; strides {token, head} in elements, innermost dim contiguous.
 * k_buffer / v_buffer may hold HBUF = 1 head shared by every kv head (the reference's MLA-style case).
 * Writes o [extend tokens][HQ][DV].  logit_cap > 0 applies cap * tanh(logit / cap).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct {
    const void *q, *k_extend, *v_extend, *k_buffer, *v_buffer;
    void* o;
    int64_t q_stride[2], k_extend_stride[2], v_extend_stride[2], k_buffer_stride[2], v_buffer_stride[2], o_stride[2];
    const void* req_to_tokens;       /* [B][L] int32 or int64 */
    int64_t req_to_tokens_stride;
    int32_t req_to_tokens_is64;
    const int64_t* b_req_idx;        /* [B] */
    const int64_t* b_seq_len;        /* [B] */
    const int32_t* b_seq_len_extend; /* [B] */
    const int32_t* b_start_loc_extend; /* [B] */
    int32_t B, HQ, HKV, HBUF, D, DV, max_len_extend;
    float sm_scale, logit_cap;
} sglk_extend_attention_args;

int sglk_extend_attention(const sglk_extend_attention_args* args, void* stream);

/* flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, is_causal)
 *                             (/root/reference/test_flash_attn_varlen.py:100-108; oracle flash_attn_varlen_ref :14-46)
 * q [Tq][HQ][D], k [Tk][HKV][D], v [Tk][HKV][DV], o [Tq][HQ][DV] bf16, strides in elements; sequence b = rows
 * cu_seqlens_q[b]..cu_seqlens_q[b+1] of q and cu_seqlens_k[b]..cu_seqlens_k[b+1] of k, v (int32, B+1 entries);
 * causal = the top-left aligned mask of scaled_dot_product_attention(is_causal=True): query i sees keys 0..i.
 * D multiple of 8, DV even, both <= 128 (72 / 80 / 94 of the reference's cases run zero-padded). */
typedef struct {
    const void* q; int64_t q_stride[2];
    const void* k; int64_t k_stride[2];
    const void* v; int64_t v_stride[2];
    void* o; int64_t o_stride[2];
    const int32_t* cu_seqlens_q;
    const int32_t* cu_seqlens_k;
    int32_t B, max_seqlen_q, HQ, HKV, D, DV;
    int32_t causal;
    float sm_scale;
} sglk_flash_attn_varlen_args;
int sglk_flash_attn_varlen(const sglk_flash_attn_varlen_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * decode_attention            replaces torch.ops.sgl_kernel.decode_attention_cpu
 *                             (/root/reference/test_mla.py:115-128, test_decoding.py:107-120)
 * 1. k_buffer[loc[b]] = key[b]; v_buffer[loc[b]] = value[b] (bit-exact), 2. one-token attention of q[b] over
 * req_to_token[b_req_idx[b]][0 .. b_seq_len[b]) in `splits` key ranges whose partial results (O/l, log-sum-exp)
 * go to the caller's scratch attn_logits [B][HQ][splits][DV+1] f32, 3. merge into o [B][HQ][DV] bf16.
 * v_buffer may be a view of k_buffer (MLA: v = k[..., :DV]); then K is read once and serves as V.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct {
    const void* q;
    void *k_buffer, *v_buffer, *o;
    const void *key, *value;
    int64_t q_stride[2], k_buffer_stride[2], v_buffer_stride[2], o_stride[2], key_stride[2], value_stride[2];
    const void* loc;                 /* [B] int32 or int64 */
    int32_t loc_is64;
    float* attn_logits;
    const void* req_to_token;        /* [B][L] int32 or int64 */
    int64_t req_to_token_stride;
    int32_t req_to_token_is64;
    const int64_t* b_req_idx;
    const int64_t* b_seq_len;
    int32_t B, HQ, HKV, D, DV, splits;
    float sm_scale, logit_cap;
} sglk_decode_attention_args;

int sglk_decode_attention(const sglk_decode_attention_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Stages of fused_experts exposed for tests and profiling (same kernels the fused call launches).
 * moe_align: counting sort of the M*topk slots by expert (ids outside [0,E) dropped), stable in slot order.
 *   sorted_slot [M*topk] i32   slot = m*topk + j, grouped by expert
 *   expert_off  [E+1]    i32   exclusive prefix of per-expert counts
 *   tile_info   [max_tiles][4] i32  {expert, first position, rows (<= tile_m), 0}; num_tiles[0] = count
 * --------------------------------------------------------------------------------------------------------- */
size_t sglk_moe_align_workspace_bytes(int32_t M, int32_t E, int32_t topk);
int32_t sglk_moe_max_tiles(int32_t M, int32_t E, int32_t topk, int32_t tile_m);
int sglk_moe_align(const int32_t* topk_ids, int32_t M, int32_t E, int32_t topk, int32_t tile_m,
                   int32_t* sorted_slot, int32_t* expert_off, int32_t* tile_info, int32_t* num_tiles,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Stage timer (measurement only): a pool of HIP events recorded on the caller's stream between the stages of
 * fused_experts, so bench.py can report the average duration of each kernel of the timed region itself
 * (roofline.achieved) without a profiler.  Stages: 0 align, 1 gemm1 (gate/up + SiLU*mul), 2 gemm2, 3 combine.
 * create(max_calls) -> handle; pass it in args.stage_timer; read() synchronises on the last event and returns the
 * per-stage mean in milliseconds over the calls recorded since the last reset.
 * --------------------------------------------------------------------------------------------------------- */
#define SGLK_NUM_STAGES 4
void* sglk_stage_timer_create(int32_t max_calls);
void sglk_stage_timer_destroy(void* timer);
void sglk_stage_timer_reset(void* timer);
int sglk_stage_timer_read(void* timer, float* mean_ms /* [SGLK_NUM_STAGES] host */, int32_t* calls /* host */);

#ifdef __cplusplus
}
#endif
#endif /* SGLK_H_ */
