// Probe: semantics of direct-to-LDS 16-byte loads on gfx950 (lane i of a wave lands at M0 base + 16*i).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k1(const u32x4* __restrict__ g, u32x4* out) {
  __shared__ __attribute__((aligned(16))) u32x4 buf[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + blockIdx.x * 256 + wave * 64 + (lane ^ 5)),
                                   (__attribute__((address_space(3))) void*)(buf + wave * 64), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = buf[threadIdx.x];
}
__global__ void k2(const u32x4* __restrict__ g, u32x4* out, int bytes) {
  __shared__ __attribute__((aligned(16))) u32x4 buf[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(buf + wave * 64), 16,
                                           (blockIdx.x * 256 + wave * 64 + (lane ^ 5)) * 16, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = buf[threadIdx.x];
}
int main() {
  const int n = 4 * 256;
  std::vector<u32x4> h(n);
  for (int i = 0; i < n; ++i) h[i] = u32x4{(unsigned)i, (unsigned)i + 1000, (unsigned)i + 2000, (unsigned)i + 3000};
  u32x4 *g, *o;
  hipMalloc(&g, n * 16); hipMalloc(&o, n * 16);
  hipMemcpy(g, h.data(), n * 16, hipMemcpyHostToDevice);
  for (int which = 0; which < 2; ++which) {
    hipMemset(o, 0, n * 16);
    if (which == 0) hipLaunchKernelGGL(k1, dim3(4), dim3(256), 0, 0, g, o);
    else hipLaunchKernelGGL(k2, dim3(4), dim3(256), 0, 0, g, o, (n - 256 + 128) * 16);   // last half block out of bounds -> zeros
    std::vector<u32x4> r(n);
    hipMemcpy(r.data(), o, n * 16, hipMemcpyDeviceToHost);
    int bad = 0, zero = 0;
    for (int i = 0; i < n; ++i) {
      int base = i & ~63, lane = i & 63, src = base + (lane ^ 5);
      bool oob = which == 1 && src >= n - 256 + 128;
      unsigned want = oob ? 0u : (unsigned)src;
      if (r[i][0] != want || r[i][3] != (oob ? 0u : want + 3000)) ++bad;
      if (oob) ++zero;
    }
    printf("kernel %d: %d mismatches (%d out-of-bounds granules expected zero)\n", which, bad, zero);
  }
  return 0;
}
