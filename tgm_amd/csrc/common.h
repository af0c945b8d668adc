// Shared host/device helpers for libtgm_amd.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tgm_amd.h"

namespace tgmx {

constexpr int kWave = 64;  // CDNA wavefront width

void set_error(const char* fmt, ...);

#define TGMX_REQUIRE(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      ::tgmx::set_error(__VA_ARGS__);    \
      return TGMX_E_INVALID;             \
    }                                    \
  } while (0)

#define TGMX_CHECK_LAUNCH(what)                                                        \
  do {                                                                                 \
    hipError_t e__ = hipGetLastError();                                                \
    if (e__ != hipSuccess) {                                                           \
      ::tgmx::set_error("%s: launch failed: %s", what, hipGetErrorString(e__));        \
      return TGMX_E_LAUNCH;                                                            \
    }                                                                                  \
  } while (0)

// q = n / d for n < 2^32 / d via one v_mul_hi_u32 (m = floor(2^32 / d) + 1).
struct FastDiv {
  uint32_t d, m;
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return d <= 1 ? n : __umulhi(n, m); }
};
inline FastDiv make_fastdiv(uint32_t div) {
  FastDiv f;
  f.d = div;
  f.m = div <= 1 ? 0u : (uint32_t)((1ull << 32) / div) + 1u;
  return f;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

}  // namespace tgmx
