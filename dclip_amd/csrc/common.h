// Shared host/device helpers for libdclip_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/dclip_hip.h"

#define DCLIP_API extern "C" __attribute__((visibility("default")))

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void dclip_set_error(const char* fmt, ...);

#define DCLIP_REQUIRE(cond, ...)                         \
  do {                                                   \
    if (!(cond)) {                                       \
      dclip_set_error(__VA_ARGS__);                      \
      return DCLIP_EINVAL;                               \
    }                                                    \
  } while (0)

// Launch check that does not synchronise: hipGetLastError only reports launch-time failures.
#define DCLIP_CHECK_LAUNCH(name)                                                      \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess) {                                                          \
      dclip_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));         \
      return DCLIP_ELAUNCH;                                                           \
    }                                                                                 \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }

// ---- wave64 reductions (DPP/permute based shuffles; the wave is 64 lanes on gfx950) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// reduce over the 32 lanes that share (lane >> 5)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float half_max(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// reduce over the 16 lanes that share (lane >> 4)
__device__ __forceinline__ float quarter_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float quarter_max(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// x * sigmoid(1.702 x) (hf:activations.py:122-123).  The sigmoid's reciprocal is v_rcp_f32 (1 ulp): the IEEE division
// sequence is ~10 instructions per element, and an epilogue wave beside MFMA-issuing waves gets one instruction through
// every ~45 cycles (tools/gemm_stamps.py) — instruction count is what the epilogue costs.
__device__ __forceinline__ float quick_gelu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float quick_gelu_grad_f(float x) {
  float s = __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
  return s * (1.0f + 1.702f * x * (1.0f - s));
}
