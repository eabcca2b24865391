// Shared helpers for the deeprl_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/deeprl_amd.h"  // exported signatures are checked against the public header

#define DRA_OK 0
#define DRA_EINVAL (-22)
#define DRA_ENOMEM (-12)

// Every export returns 0 or an error code; nothing throws or aborts across the C ABI.
#define DRA_HIP(expr)                                  \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) return (int)_e;              \
  } while (0)

#define DRA_LAUNCH_CHECK()                             \
  do {                                                 \
    hipError_t _e = hipGetLastError();                 \
    if (_e != hipSuccess) return (int)_e;              \
  } while (0)

#define DRA_API extern "C" __attribute__((visibility("default")))

static inline hipStream_t dra_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;  // CDNA4 wavefront

// 64-lane butterfly reductions (wave64: offsets up to 32).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
