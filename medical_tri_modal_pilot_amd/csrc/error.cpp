// Thread-local last-error string of the C-ABI (include/mtmp.h: mtmp_last_error).
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void mtmp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mtmp_last_error(void) { return g_err; }
extern "C" int mtmp_abi_version(void) { return 6; }
