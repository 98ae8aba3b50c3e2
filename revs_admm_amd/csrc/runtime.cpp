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
