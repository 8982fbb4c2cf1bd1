// Version and error reporting of the C ABI (see include/tst_hip.h).
#include "tst_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void tst_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int tst_version(void) { return 109; }

extern "C" const char* tst_last_error(void) { return g_err; }
