// Error bookkeeping + version for libsempyr_hip.so
#include <stdarg.h>
#include <stdio.h>
#include "../../include/sempyr.h"

static thread_local char g_err[512] = "";

void sempyr_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int sempyr_version(void) { return SEMPYR_ABI_VERSION; }
extern "C" const char* sempyr_last_error_string(void) { return g_err; }
