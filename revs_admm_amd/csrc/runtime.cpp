// Error reporting and version string of librevs_admm.so.
#include "common.h"
#include <stdarg.h>

namespace revs {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace revs

extern "C" const char *revs_last_error(void) { return revs::g_err; }
extern "C" const char *revs_version(void) { return "revs_admm_amd 0.1 (gfx950)"; }

// Device-side address of pinned host memory (hipHostMalloc / torch pin_memory): lets a
// kernel write its few result words where the host reads them, without a copy kernel.
extern "C" int revs_host_device_ptr(void *host_ptr, void **dev_ptr) {
    REVS_REQUIRE(host_ptr && dev_ptr, "revs_host_device_ptr: null argument");
    const hipError_t e = hipHostGetDevicePointer(dev_ptr, host_ptr, 0);
    if (e != hipSuccess) {
        revs::set_error("revs_host_device_ptr: %s", hipGetErrorString(e));
        return REVS_EINVAL;
    }
    return REVS_OK;
}
