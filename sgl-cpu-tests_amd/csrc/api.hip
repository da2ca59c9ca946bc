// Version, error string and device queries of the sglk C-ABI (include/sglk.h).
#include <stdarg.h>
#include <string.h>

#include <stdlib.h>

#include <mutex>

#include "knobs.h"

namespace sglk {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
Knobs g_knobs;
std::once_flag g_knobs_once;

int env_int(const char* name, int unset) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : unset;
}
bool env_set(const char* name) { return getenv(name) != nullptr; }

void read_env() {
    Knobs k;
    k.moe_tile_m = env_int("SGLK_MOE_TILE_M", 0);
    k.mid_lo = env_int("SGLK_MID_LO", 8);
    k.mid_hi = env_int("SGLK_MID_HI", 0);
    k.dense_mid_max = env_int("SGLK_DENSE_MID_MAX", 2048);
    k.shared_mid_max = env_int("SGLK_SHARED_MID_MAX", 1024);
    k.shared_i8_mid_max = env_int("SGLK_SHARED_I8_MID_MAX", 1024);
    k.shared_big_wgs = env_int("SGLK_SHARED_BIG_WGS", 64);
    k.no_tuned_splitk = env_set("SGLK_NO_TUNED_SPLITK");
    k.mid_i8_hi = env_int("SGLK_MID_I8_HI", 44);
    k.mid_bf16_hi = env_int("SGLK_MID_BF16_HI", 44);
    k.force_generic = env_set("SGLK_FORCE_GENERIC");
    k.no_i8_mid = env_set("SGLK_NO_I8_MID");
    k.i8_dense_mid_wgs = env_int("SGLK_I8_DENSE_MID_WGS", 64);
    k.bf16_mid_target = env_int("SGLK_BF16_MID_TARGET", 0);
    k.mid_dense_model = env_int("SGLK_MID_DENSE_MODEL", 2);
    k.dense_mid_wgs_bf16 = env_int("SGLK_DENSE_MID_WGS_BF16", 60);
    k.dense_mid_wgs_fp8 = env_int("SGLK_DENSE_MID_WGS_FP8", 96);
    k.no_bf16_mid = env_set("SGLK_NO_BF16_MID");
    k.tail_split = env_int("SGLK_TAIL_SPLIT", -1);
    k.mid_down2 = env_int("SGLK_MID_DOWN2", -1);
    k.bf16_w4 = env_set("SGLK_BF16_W4");
    k.align_3pass = env_set("SGLK_ALIGN_3PASS");
    k.wide_n = env_set("SGLK_WIDE_N");
    k.persist = env_int("SGLK_PERSIST", -1);
    k.max_wgs = env_int("SGLK_MAX_WGS", 0);
    k.dec_splits = env_int("SGLK_DEC_SPLITS", 0);
    k.dec_nt = env_int("SGLK_DEC_NT", -1);
    k.dec_fold = env_int("SGLK_DEC_FOLD", -1);
    k.w_nt = env_int("SGLK_W_NT", -1);
    k.attn_nw = env_int("SGLK_ATTN_NW", 0);
    k.attn_order = env_int("SGLK_ATTN_ORDER", -1);
    k.attn_pair = env_int("SGLK_ATTN_PAIR", -1);
    k.attn_pp = env_int("SGLK_ATTN_PP", -1);
    k.mxfp4_native = env_int("SGLK_MXFP4_NATIVE", -1);
    k.mxfp4_rt = env_int("SGLK_MXFP4_RT", 4);
    k.no_pack_on_the_fly = env_set("SGLK_NO_PACK_ON_THE_FLY");
    k.pack_min_rows = env_int("SGLK_PACK_MIN_ROWS", 0);
    k.inline_align_max = env_int("SGLK_INLINE_ALIGN_MAX", 16);
    k.no_block_fold = env_set("SGLK_NO_BLOCK_FOLD");
    k.s128 = env_int("SGLK_S128", -1);
    k.i8_s128 = env_int("SGLK_I8_S128", -1);
    k.s128_prio = env_int("SGLK_S128_PRIO", -1);
    k.no_mid_narrow = env_set("SGLK_NO_MID_NARROW");
    k.mid_nw = env_int("SGLK_MID_NW", 0);
    k.mid_far = env_int("SGLK_MID_FAR", -1);
    k.dense_s128 = env_int("SGLK_DENSE_S128", -1);
    k.ar_wait_ms = env_int("SGLK_AR_WAIT_MS", 0);
    k.fp8_act = env_int("SGLK_FP8_ACT", 0);
    k.rescale_ablate = env_int("SGLK_RESCALE", 0);
    if (const char* dp = getenv("SGLK_DBG_PTR")) k.dbg_ptr = strtoull(dp, nullptr, 16);
    g_knobs = k;
}
}  // namespace

const Knobs& knobs() {
    std::call_once(g_knobs_once, read_env);
    return g_knobs;
}

int device_cu_count() {
    static std::atomic<int> cus[32];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return 256;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

int ensure_dyn_lds(const void* func, int bytes, std::atomic<unsigned>& done, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) SGLK_FAIL(SGLK_ERR_LAUNCH, "%s: hipGetDevice failed", what);
    const unsigned bit = 1u << (dev & 31);
    if (done.load(std::memory_order_acquire) & bit) return SGLK_OK;
    const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) SGLK_FAIL(SGLK_ERR_LAUNCH, "%s: cannot reserve %d bytes of LDS: %s", what, bytes, hipGetErrorString(e));
    done.fetch_or(bit, std::memory_order_release);
    return SGLK_OK;
}
}  // namespace sglk

extern "C" void sglk_reload_env(void) {
    sglk::knobs();        // make sure the once-flag is spent, then overwrite
    sglk::read_env();
}

extern "C" int sglk_version(void) { return SGLK_VERSION; }

extern "C" const char* sglk_last_error(void) { return sglk::g_err; }

extern "C" int sglk_device_cu_count(int dev) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        sglk::set_error("hipGetDeviceProperties(%d) failed", dev);
        return SGLK_ERR_INVALID;
    }
    return prop.multiProcessorCount;
}

extern "C" int sglk_aux_create(void** stream, void** event0, void** event1) {
    using namespace sglk;
    SGLK_REQUIRE(stream && event0 && event1, SGLK_ERR_INVALID, "aux_create: null pointer");
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // SGLK_AUX_PRIO=low|high: queue priority of the side stream (A/B knob; the tail tiles are filler work)
    int lo = 0, hi = 0;
    const char* pr = getenv("SGLK_AUX_PRIO");
    const bool want = pr && (pr[0] == 'l' || pr[0] == 'h') && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess;
    const hipError_t se = want ? hipStreamCreateWithPriority(&st, hipStreamNonBlocking, pr[0] == 'l' ? lo : hi)
                               : hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (se != hipSuccess ||
        hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess) {
        if (e0) hipEventDestroy(e0);
        if (st) hipStreamDestroy(st);
        SGLK_FAIL(SGLK_ERR_LAUNCH, "aux_create: cannot create a stream and two events");
    }
    *stream = st;
    *event0 = e0;
    *event1 = e1;
    return SGLK_OK;
}

extern "C" void sglk_aux_destroy(void* stream, void* event0, void* event1) {
    if (event0) hipEventDestroy((hipEvent_t)event0);
    if (event1) hipEventDestroy((hipEvent_t)event1);
    if (stream) hipStreamDestroy((hipStream_t)stream);
}
