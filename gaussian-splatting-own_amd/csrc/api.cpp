// Library-wide entry points of libgsr_hip.so: version and per-thread error text.
#include <cstdarg>
#include <cstdio>

#include "gsr_common.h"

namespace {
thread_local char g_error[512] = "";
}

namespace gsr {
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace gsr

extern "C" int gsr_version(void) { return GSR_VERSION; }
extern "C" const char* gsr_last_error(void) { return g_error; }
