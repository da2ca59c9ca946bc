// Version, error string and device queries of the sglk C-ABI (include/sglk.h).
#include <stdarg.h>
#include <string.h>

#include "sglk_common.h"

namespace sglk {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace sglk

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
