// Per-thread error string + version (libosvos_hip.so).
#include "common.h"
#include <stdlib.h>

static thread_local char g_err[512] = "";

void osvos_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* osvos_last_error(void) { return g_err; }
extern "C" int osvos_version(void) { return OSVOS_ABI_VERSION; }
