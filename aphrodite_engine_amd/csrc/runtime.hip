// Error reporting + ABI version for libaphrodite_mi355x.so
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace aphro {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace aphro

extern "C" const char* aphro_last_error(void) { return aphro::g_err; }
extern "C" int aphro_abi_version(void) { return 1; }
