// Shared host-side plumbing of libgsr_hip.so: error reporting and launch checks.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../include/gsr.h"
#include "../../include/gsr_constants.h"

namespace gsr {

// Records the message returned by gsr_last_error() for this thread; returns code.
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GSR_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return GSR_OK;
}

inline int div_up(int a, int b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace gsr

#define GSR_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return gsr::fail(GSR_E_INVALID, __VA_ARGS__);   \
  } while (0)
