// Direct all-reduce between the GPUs of one node over xGMI peer memory (replaces the role of the reference's shared-memory
// all-reduce between CPU ranks, /root/reference/test_allreduce.py:86-105; its bench message is 1024 x 5120 bf16 = 10 MiB,
// run_allreduce_cpu.sh:8-12).
//
// MI355X's eight GPUs are fully connected by point-to-point links (7 x ~153 GB/s per GPU), so a ring is bound by ONE link while
// direct reads use all of them at once.  Every rank owns a staging region that its peers map through HIP IPC:
//   one-shot  (small messages): every rank reads all G staged copies and sums them in rank order       -- (G-1) x n bytes in
//   two-shot  (large messages): rank r sums slice r of all copies (reduce-scatter), publishes it, and every rank gathers the
//                               G reduced slices (all-gather)                                  -- 2 (G-1)/G x n bytes in
// Sums are fp32 in ascending rank order with one rounding, so every rank computes bit-identical results, run to run.
//
// Synchronisation (robust rather than minimal; the payload, not the handshake, sets the time at these sizes):
//   * a step's data is complete when the KERNEL that wrote it has ended (kernel boundaries write the L2 back), and only then
//     a one-thread kernel stores the step's epoch number into every peer's flag word (system-scope atomic stores into
//     fine-grained memory);
//   * consumers poll their OWN flag words (system-scope loads) with a bounded spin -- a peer that never arrives sets a status
//     word and the kernel exits instead of hanging the GPU -- and read peer data with system-scope (sc0 sc1) loads;
//   * staging is double-buffered by epoch parity: a rank can only start call e+2 after its peers signalled call e+1, which
//     they do after finishing call e -- so nobody overwrites a buffer that is still being read, without a trailing barrier.
#include <string.h>

#include "knobs.h"
#include "sglk_common.h"

namespace sglk {

constexpr int kMaxRanks = 8;
// wall_clock64() runs at 100 MHz.  A consumer gives up on a peer after knobs().ar_wait_ms (SGLK_AR_WAIT_MS, default 30 s: a first
// call's warm-up, a garbage-collection pause or a slow peer are not failures; a collective library would wait for ever, a
// kernel must not), sets the status word and leaves `out` unwritten: the caller sees the status at its next call
// (collectives.py) and has to resynchronise the group before using the communicator again.

struct CommView {
    unsigned char* data[kMaxRanks];   // each rank's staging region: [2 parities][2 areas: input copy, reduced slices][cap bytes]
    unsigned* flags[kMaxRanks];       // each rank's flag words: [2 phases][kMaxRanks]
    int rank, world;
    long long cap;                    // bytes per area
};

// 16 bytes at byte offset `off` of a peer's area, at SYSTEM scope (sc0 sc1: bypasses this GPU's caches -- the line lives in
// another GPU's memory and is rewritten every call); the compiler tracks the load like any other, so several stay in flight
SGLK_DEV u32x4 ld_sys(const unsigned char* area, unsigned bytes, unsigned off) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)area, 0, bytes, 0x00020000);
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 17));
}

__global__ __launch_bounds__(256) void ar_copy_kernel(const uint4* __restrict__ in, uint4* __restrict__ dst, long long n16) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) dst[i] = in[i];
    // the staging area is ordinary (coarse-grained) device memory that PEERS read: make this thread's stores visible at system
    // scope before the kernel ends, rather than relying on the end-of-kernel write-back alone
    __threadfence_system();
}

// one thread: tell every rank (myself included) that my step `phase` of call `epoch` is complete
__global__ void ar_signal_kernel(CommView c, int phase, unsigned epoch) {
    if (threadIdx.x < (unsigned)c.world)
        __hip_atomic_store(c.flags[threadIdx.x] + phase * kMaxRanks + c.rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// every rank's flag for `phase` has reached `epoch` (one lane per workgroup polls; bounded)
SGLK_DEV bool wait_all(const CommView& c, int phase, unsigned epoch, int* status, long long wait_ticks) {
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        int ok = 1;
        for (int r = 0; r < c.world && ok; ++r) {
            const unsigned* f = c.flags[c.rank] + phase * kMaxRanks + r;
            const long long t0 = wall_clock64();
            // epochs only grow; the subtraction handles wrap-around
            while ((int)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
                if (wall_clock64() - t0 > wait_ticks) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        if (!ok) atomicExch(status, 1);
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}

SGLK_DEV void add8(float* acc, const u32x4& v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc[2 * q] += __uint_as_float(v[q] << 16);
        acc[2 * q + 1] += __uint_as_float(v[q] & 0xffff0000u);
    }
}
SGLK_DEV u32x4 pack8(const float* acc) {
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = pack_bf16x2(acc[2 * q], acc[2 * q + 1]);
    return o;
}

// out[i] = bf16(sum over ranks r ascending of copy_r[i]) for 16-byte chunks [c0, c1)
__global__ __launch_bounds__(256) void ar_reduce_kernel(CommView c, long long area_off, long long c0, long long c1,
                                                        uint4* __restrict__ out, long long out_c0, unsigned epoch, int* status,
                                                        long long wait_ticks) {
    if (!wait_all(c, 0, epoch, status, wait_ticks)) return;
    for (long long i = c0 + (long long)blockIdx.x * 256 + threadIdx.x; i < c1; i += (long long)gridDim.x * 256) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        u32x4 v[kMaxRanks];
#pragma unroll
        for (int r = 0; r < kMaxRanks; ++r)
            if (r < c.world) v[r] = ld_sys(c.data[r] + area_off, (unsigned)c.cap, (unsigned)(i * 16));
#pragma unroll
        for (int r = 0; r < kMaxRanks; ++r)
            if (r < c.world) add8(acc, v[r]);
        const u32x4 o = pack8(acc);
        out[i - c0 + out_c0] = make_uint4(o[0], o[1], o[2], o[3]);
    }
    __threadfence_system();   // two-shot: `out` is my reduced area, which the peers gather (see ar_copy_kernel)
}

// two-shot, second half: out[slice r] = rank r's reduced slice
__global__ __launch_bounds__(256) void ar_gather_kernel(CommView c, long long area_off, long long per, long long n16,
                                                        uint4* __restrict__ out, unsigned epoch, int* status, long long wait_ticks) {
    if (!wait_all(c, 1, epoch, status, wait_ticks)) return;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / per);
        const u32x4 v = ld_sys(c.data[r] + area_off, (unsigned)c.cap, (unsigned)((i - (long long)r * per) * 16));
        out[i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace sglk

using namespace sglk;

extern "C" int sglk_ipc_export(const void* dev_ptr, void* handle64) {
    SGLK_REQUIRE(dev_ptr && handle64, SGLK_ERR_INVALID, "ipc_export: null pointer");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    const hipError_t e = hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, (void*)dev_ptr);
    if (e != hipSuccess) SGLK_FAIL(SGLK_ERR_LAUNCH, "ipc_export: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", hipGetErrorString(e));
    return SGLK_OK;
}

extern "C" int sglk_ipc_open(const void* handle64, void** dev_ptr) {
    SGLK_REQUIRE(dev_ptr && handle64, SGLK_ERR_INVALID, "ipc_open: null pointer");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    const hipError_t e = hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) SGLK_FAIL(SGLK_ERR_LAUNCH, "ipc_open: %s", hipGetErrorString(e));
    return SGLK_OK;
}

extern "C" int sglk_ipc_close(void* dev_ptr) {
    if (dev_ptr && hipIpcCloseMemHandle(dev_ptr) != hipSuccess) SGLK_FAIL(SGLK_ERR_LAUNCH, "ipc_close failed");
    return SGLK_OK;
}

// Device memory for a communicator, zero-filled, as its OWN allocation (an IPC handle names a whole allocation, so the staging
// region must not be a slice of somebody's pool): finegrained != 0 -> uncached fine-grained memory for the flag words
extern "C" int sglk_comm_alloc(size_t bytes, int32_t finegrained, void** dev_ptr) {
    SGLK_REQUIRE(dev_ptr && bytes > 0, SGLK_ERR_INVALID, "comm_alloc: bad arguments");
    void* p = nullptr;
    const hipError_t e = finegrained ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) : hipMalloc(&p, bytes);
    if (e != hipSuccess || !p) SGLK_FAIL(SGLK_ERR_LAUNCH, "comm_alloc: %s", hipGetErrorString(e));
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        hipFree(p);
        SGLK_FAIL(SGLK_ERR_LAUNCH, "comm_alloc: memset failed");
    }
    *dev_ptr = p;
    return SGLK_OK;
}

extern "C" void sglk_comm_free(void* dev_ptr) {
    if (dev_ptr) hipFree(dev_ptr);
}

extern "C" int sglk_allreduce_sum_bf16(void* const* peer_data, void* const* peer_flags, int32_t rank, int32_t world,
                                       int64_t capacity_bytes, const void* in, void* out, int64_t n_elems, uint32_t epoch,
                                       int32_t algo, int32_t* status_dev, void* stream) {
    SGLK_REQUIRE(peer_data && peer_flags && world >= 1 && world <= kMaxRanks && rank >= 0 && rank < world, SGLK_ERR_INVALID,
                 "allreduce: bad communicator (world %d, rank %d)", world, rank);
    SGLK_REQUIRE(n_elems >= 0 && n_elems % 8 == 0 && n_elems * 2 <= capacity_bytes && capacity_bytes % 256 == 0 &&
                     capacity_bytes < (1ll << 31), SGLK_ERR_SHAPE,
                 "allreduce: %lld bf16 elements do not fit the staging capacity (%lld bytes) or are not a multiple of 8",
                 (long long)n_elems, (long long)capacity_bytes);
    SGLK_REQUIRE(status_dev && (n_elems == 0 || (in && out)), SGLK_ERR_INVALID, "allreduce: null pointer");
    SGLK_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0, SGLK_ERR_SHAPE, "allreduce: in / out must be 16-byte aligned");
    if (n_elems == 0) return SGLK_OK;
    hipStream_t s = (hipStream_t)stream;
    CommView c{};
    for (int r = 0; r < world; ++r) {
        SGLK_REQUIRE(peer_data[r] && peer_flags[r], SGLK_ERR_INVALID, "allreduce: peer %d not mapped", r);
        c.data[r] = (unsigned char*)peer_data[r];
        c.flags[r] = (unsigned*)peer_flags[r];
    }
    c.rank = rank;
    c.world = world;
    c.cap = capacity_bytes;
    const long long n16 = n_elems / 8;
    const long long par = epoch & 1;
    const long long in_area = par * 2 * c.cap, red_area = in_area + c.cap;
    const int blocks = (int)(n16 < 256 * 512 ? ceil_div(n16, 256) : 512);   // never the whole chip: a polling kernel must not
                                                                              // starve the kernels it is waiting for
    // two-shot when slices are whole 16-byte chunks and the message is large (algo: 0 = by size, 1 = one-shot, 2 = two-shot)
    const bool two = algo == 2 || (algo == 0 && world > 2 && n_elems * 2 >= (1 << 20));
    const long long wait_ticks = (long long)(knobs().ar_wait_ms > 0 ? knobs().ar_wait_ms : 30000) * 100000ll;
    hipLaunchKernelGGL(ar_copy_kernel, dim3(blocks), dim3(256), 0, s, (const uint4*)in, (uint4*)(c.data[rank] + in_area), n16);
    hipLaunchKernelGGL(ar_signal_kernel, dim3(1), dim3(64), 0, s, c, 0, epoch);
    if (!two) {
        hipLaunchKernelGGL(ar_reduce_kernel, dim3(blocks), dim3(256), 0, s, c, in_area, 0ll, n16, (uint4*)out, 0ll, epoch, status_dev, wait_ticks);
    } else {
        const long long per = ceil_div(n16, world);
        const long long c0 = (long long)rank * per < n16 ? (long long)rank * per : n16;
        const long long c1 = c0 + per < n16 ? c0 + per : n16;
        // my slice of the sum -> my reduced area (position 0 of it), then publish; then gather everybody's slice
        hipLaunchKernelGGL(ar_reduce_kernel, dim3(blocks), dim3(256), 0, s, c, in_area, c0, c1, (uint4*)(c.data[rank] + red_area), 0ll,
                           epoch, status_dev, wait_ticks);
        hipLaunchKernelGGL(ar_signal_kernel, dim3(1), dim3(64), 0, s, c, 1, epoch);
        hipLaunchKernelGGL(ar_gather_kernel, dim3(blocks), dim3(256), 0, s, c, red_area, per, n16, (uint4*)out, epoch, status_dev, wait_ticks);
    }
    SGLK_CHECK_LAUNCH("allreduce");
    return SGLK_OK;
}
