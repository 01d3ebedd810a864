// common.h — shared host/device helpers for liblnrf (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/lnrf.h"

namespace lnrf {

void set_error(const char* fmt, ...);

inline int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s", what, hipGetErrorString(e));
  return (int)e;
}

#define LNRF_CHECK_ARG(cond, msg)                 \
  do {                                            \
    if (!(cond)) {                                \
      lnrf::set_error("%s: %s", __func__, msg);   \
      return LNRF_ERR_ARG;                        \
    }                                             \
  } while (0)

#define LNRF_LAUNCH_CHECK()                                    \
  do {                                                         \
    hipError_t e_ = hipGetLastError();                         \
    if (e_ != hipSuccess) return lnrf::hip_fail(e_, __func__); \
  } while (0)

static inline hipStream_t as_stream(lnrf_stream_t s) { return (hipStream_t)s; }

// A/B switches of past experiments exist in debug builds only (-DLNRF_EXPERIMENTS,
// tools/build_experiments.sh): the product library reads no environment variable and always takes the default.
#ifdef LNRF_EXPERIMENTS
inline int exp_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
inline bool exp_env_is(const char* name, char first) {
  const char* v = getenv(name);
  return v && v[0] == first;
}
#else
inline int exp_env_int(const char*, int dflt) { return dflt; }
inline bool exp_env_is(const char*, char) { return false; }
#endif

constexpr int kWave = 64;

// ---- wave64 helpers -------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    float t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}

}  // namespace lnrf
