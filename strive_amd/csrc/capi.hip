// Error reporting + version of the C ABI.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void strive_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* strive_last_error(void) { return g_err; }
extern "C" int strive_abi_version(void) { return STRIVE_ABI_VERSION; }
