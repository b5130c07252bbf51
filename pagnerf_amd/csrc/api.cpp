// ABI version + thread-local error string.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/pagnerf_hip.h"

static thread_local char g_err[512] = "";

void pag_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int pag_abi_version(void) { return PAG_ABI_VERSION; }
extern "C" const char *pag_last_error_string(void) { return g_err; }
