// Shared host/device helpers for libmmd_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

struct mmd_profiler_s;

namespace mmd {

void set_error(const char* fmt, ...);

#define MMD_HIP_CHECK(expr)                                                                      \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      mmd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return 1;                                                                                  \
    }                                                                                            \
  } while (0)

#define MMD_REQUIRE(cond, ...)      \
  do {                              \
    if (!(cond)) {                  \
      mmd::set_error(__VA_ARGS__);  \
      return 2;                     \
    }                               \
  } while (0)

// measurement brackets of include/mmd_amd_debug.h (unet.hip): true = an event was recorded and prof_end must follow
bool prof_begin(::mmd_profiler_s* prof, int counter, int kind, hipStream_t st);
void prof_end(::mmd_profiler_s* prof, hipStream_t st);
void prof_skip(::mmd_profiler_s* prof, int counter);   // a launch that did not happen (fused into another): keeps the counters in step

constexpr int H = 64;   // support points per trajectory
constexpr int D = 4;    // state dim (x, y, vx, vy)

}  // namespace mmd
