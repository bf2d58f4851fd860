// Host-side plumbing of libicrl_a2c.so: error text and ABI version.
#include <stdarg.h>
#include <stdio.h>

#include "icrl_common.h"

namespace icrl {
static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}
}  // namespace icrl

extern "C" int icrl_abi_version(void) { return ICRL_ABI_VERSION; }
extern "C" const char* icrl_last_error(void) { return icrl::g_error; }
