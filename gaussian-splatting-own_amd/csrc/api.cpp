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

// Host-side timing events for the *_timed entry points (hipEvent_t handed around as void*).
extern "C" void* gsr_event_create(void) {
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) {
    gsr::fail(GSR_E_LAUNCH, "event_create: %s", hipGetErrorString(hipGetLastError()));
    return nullptr;
  }
  return e;
}

extern "C" void* gsr_event_create_untimed(void) {
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
    gsr::fail(GSR_E_LAUNCH, "event_create_untimed: %s", hipGetErrorString(hipGetLastError()));
    return nullptr;
  }
  return e;
}

extern "C" void gsr_event_destroy(void* event) {
  if (event) (void)hipEventDestroy((hipEvent_t)event);
}

extern "C" int gsr_event_elapsed_ms(void* start_event, void* stop_event, float* h_ms) {
  GSR_REQUIRE(start_event && stop_event && h_ms, "event_elapsed_ms: null pointer");
  hipError_t e = hipEventSynchronize((hipEvent_t)stop_event);
  if (e == hipSuccess) e = hipEventElapsedTime(h_ms, (hipEvent_t)start_event, (hipEvent_t)stop_event);
  if (e != hipSuccess) return gsr::fail(GSR_E_LAUNCH, "event_elapsed_ms: %s", hipGetErrorString(e));
  return GSR_OK;
}
