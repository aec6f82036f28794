// Error reporting and build identification of the C-ABI library.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

int bigc_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* bigc_last_error(void) { return g_err; }
extern "C" const char* bigc_build_info(void) { return "bigc_hip gfx950 f32-mfma " __DATE__ " " __TIME__; }
