// Library-level entry points: version + thread-local error string.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace tgmx {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace tgmx

extern "C" int tgmx_version(void) { return TGMX_ABI_VERSION; }
extern "C" const char* tgmx_last_error(void) { return tgmx::g_err; }
