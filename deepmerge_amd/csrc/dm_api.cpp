// Library-level entry points and the thread-local error string of libdeepmerge_hip.
#include <cstdarg>
#include <cstdio>

#include "deepmerge_hip.h"

static thread_local char g_err[512] = "";

void dm_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int dm_abi_version(void) { return 6; }
extern "C" const char *dm_last_error(void) { return g_err; }
extern "C" const char *dm_arch(void) { return "gfx950"; }
