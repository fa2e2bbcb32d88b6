// Error reporting + tiny elementwise helpers.
#include "common.h"

static thread_local char g_err[512] = "";

void dclip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

DCLIP_API int dclip_abi_version(void) { return DCLIP_ABI_VERSION; }
DCLIP_API const char* dclip_last_error(void) { return g_err; }

namespace {

__global__ void __launch_bounds__(256) axpby_kernel(const float* __restrict__ x, float* __restrict__ y, float a,
                                                    float b, size_t n4, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 yv = (b != 0.f) ? reinterpret_cast<f32x4*>(y)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    reinterpret_cast<f32x4*>(y)[i] = a * xv + b * yv;
  }
  // tail (n not a multiple of 4)
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = a * x[i] + (b != 0.f ? b * y[i] : 0.f);
}

__global__ void __launch_bounds__(256) fill_kernel(float* __restrict__ y, float v, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = v;
}

inline int grid_for(size_t work) {
  size_t b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

DCLIP_API int dclip_axpby(const float* x, float* y, float a, float b, size_t n, void* stream) {
  DCLIP_REQUIRE(x && y, "axpby: null pointer");
  if (n == 0) return DCLIP_OK;
  const bool vec = (((uintptr_t)x | (uintptr_t)y) % 16) == 0;
  const size_t n4 = vec ? n / 4 : 0;
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n4 ? n4 : n)), dim3(256), 0, (hipStream_t)stream, x, y, a, b, n4, n);
  DCLIP_CHECK_LAUNCH("axpby");
  return DCLIP_OK;
}

DCLIP_API int dclip_fill(float* y, float v, size_t n, void* stream) {
  DCLIP_REQUIRE(y, "fill: null pointer");
  if (n == 0) return DCLIP_OK;
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, v, n);
  DCLIP_CHECK_LAUNCH("fill");
  return DCLIP_OK;
}
