// Row-wise pieces of the losses: F.normalize forward/backward, the cosine distillation loss and
// deterministic scalar reductions.  One wave per embedding row; reductions by wave shuffles.
// (The similarity matrix itself lives in gemm_f32.hip: dclip_contrastive_lse / _grad.)
#include "common.h"

namespace {

// xhat = x / max(||x||, eps)
__global__ void __launch_bounds__(256) normalize_fwd_kernel(const float* __restrict__ x, float* __restrict__ xhat,
                                                            float* __restrict__ inv_norm, int B, int P, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * P);
  const int p4 = P >> 2;
  float s = 0.f;
  for (int i = lane; i < p4; i += 64) {
    f32x4 v = xr[i];
    s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  const float inv = 1.0f / fmaxf(sqrtf(wave_sum(s)), eps);
  if (lane == 0 && inv_norm) inv_norm[row] = inv;
  f32x4* yr = reinterpret_cast<f32x4*>(xhat + (size_t)row * P);
  for (int i = lane; i < p4; i += 64) yr[i] = xr[i] * inv;
}

// dx = inv * (dxhat - xhat <dxhat, xhat>)  when ||x|| >= eps;   dx = dxhat / eps  when the clamp was active
__global__ void __launch_bounds__(256) normalize_bwd_kernel(const float* __restrict__ dxhat, const float* __restrict__ xhat,
                                                            const float* __restrict__ inv_norm, float* __restrict__ dx,
                                                            int B, int P, float eps, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const f32x4* gr = reinterpret_cast<const f32x4*>(dxhat + (size_t)row * P);
  const f32x4* hr = reinterpret_cast<const f32x4*>(xhat + (size_t)row * P);
  const int p4 = P >> 2;
  const float inv = inv_norm[row];
  const bool clamped = inv >= 1.0f / eps;
  float dot = 0.f;
  if (!clamped) {
    for (int i = lane; i < p4; i += 64) {
      f32x4 g = gr[i], h = hr[i];
      dot += (g[0] * h[0] + g[1] * h[1]) + (g[2] * h[2] + g[3] * h[3]);
    }
    dot = wave_sum(dot);
  }
  f32x4* dr = reinterpret_cast<f32x4*>(dx + (size_t)row * P);
  for (int i = lane; i < p4; i += 64) {
    f32x4 v = (gr[i] - hr[i] * dot) * inv;
    dr[i] = accumulate ? dr[i] + v : v;
  }
}

// cos[b] = <s,t> / (max(|s|,eps) max(|t|,eps))
__global__ void __launch_bounds__(256) cosine_fwd_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                         float* __restrict__ cosv, int B, int P, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const f32x4* sr = reinterpret_cast<const f32x4*>(s + (size_t)row * P);
  const f32x4* tr = reinterpret_cast<const f32x4*>(t + (size_t)row * P);
  const int p4 = P >> 2;
  float ss = 0.f, tt = 0.f, st = 0.f;
  for (int i = lane; i < p4; i += 64) {
    f32x4 a = sr[i], b = tr[i];
    ss += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
    tt += (b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3]);
    st += (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
  }
  ss = wave_sum(ss), tt = wave_sum(tt), st = wave_sum(st);
  if (lane == 0) cosv[row] = st / (fmaxf(sqrtf(ss), eps) * fmaxf(sqrtf(tt), eps));
}

// d/ds of coef * (1 - cos):  -coef * (that - cos * shat) / max(|s|,eps)     (clamped rows: -coef * that / eps)
__global__ void __launch_bounds__(256) cosine_bwd_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                         const float* __restrict__ cosv, float* __restrict__ ds, int B,
                                                         int P, float eps, float coef, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const f32x4* sr = reinterpret_cast<const f32x4*>(s + (size_t)row * P);
  const f32x4* tr = reinterpret_cast<const f32x4*>(t + (size_t)row * P);
  const int p4 = P >> 2;
  float ss = 0.f, tt = 0.f;
  for (int i = lane; i < p4; i += 64) {
    f32x4 a = sr[i], b = tr[i];
    ss += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
    tt += (b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3]);
  }
  const float ns = sqrtf(wave_sum(ss)), nt = sqrtf(wave_sum(tt));
  const float is = 1.0f / fmaxf(ns, eps), it = 1.0f / fmaxf(nt, eps);
  const float c = (ns >= eps) ? cosv[row] : 0.f;  // clamped: shat = s/eps is linear in s, no projection term
  f32x4* dr = reinterpret_cast<f32x4*>(ds + (size_t)row * P);
  for (int i = lane; i < p4; i += 64) {
    f32x4 v = (tr[i] * it - sr[i] * (is * c)) * (-coef * is);
    dr[i] = accumulate ? dr[i] + v : v;
  }
}

// out = scale * sum_i (a[i] - b[i])   (b may be null; single workgroup, fixed order -> deterministic)
__global__ void __launch_bounds__(256) sub_reduce_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, int n, float scale, float bias,
                                                         int accumulate) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += a[i] - (b ? b[i] : 0.f);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = scale * ((red[0] + red[1]) + (red[2] + red[3])) + bias;
    *out = accumulate ? *out + v : v;
  }
}

}  // namespace

DCLIP_API int dclip_normalize_rows_fwd(const float* x, float* xhat, float* inv_norm, int B, int P, float eps,
                                       void* stream) {
  DCLIP_REQUIRE(x && xhat && B > 0 && P > 0 && P % 4 == 0, "normalize_rows_fwd: bad arguments");
  hipLaunchKernelGGL(normalize_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, x, xhat, inv_norm, B, P, eps);
  DCLIP_CHECK_LAUNCH("normalize_rows_fwd");
  return DCLIP_OK;
}

DCLIP_API int dclip_normalize_rows_bwd(const float* dxhat, const float* xhat, const float* inv_norm, float* dx, int B,
                                       int P, float eps, int accumulate, void* stream) {
  DCLIP_REQUIRE(dxhat && xhat && inv_norm && dx && B > 0 && P > 0 && P % 4 == 0, "normalize_rows_bwd: bad arguments");
  hipLaunchKernelGGL(normalize_bwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, dxhat, xhat, inv_norm, dx, B,
                     P, eps, accumulate);
  DCLIP_CHECK_LAUNCH("normalize_rows_bwd");
  return DCLIP_OK;
}

DCLIP_API int dclip_cosine_loss_fwd(const float* s, const float* t, float* loss_sum, float* cosv, int B, int P,
                                    void* stream) {
  DCLIP_REQUIRE(s && t && loss_sum && cosv && B > 0 && P > 0 && P % 4 == 0, "cosine_loss_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cosine_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, s, t, cosv, B, P, 1e-12f);
  DCLIP_CHECK_LAUNCH("cosine_loss_fwd");
  // sum_b (1 - cos_b) = B - sum cos
  hipLaunchKernelGGL(sub_reduce_kernel, dim3(1), dim3(256), 0, st, (const float*)cosv, (const float*)nullptr, loss_sum, B,
                     -1.0f, (float)B, 0);
  DCLIP_CHECK_LAUNCH("cosine_loss_fwd.reduce");
  return DCLIP_OK;
}

DCLIP_API int dclip_cosine_loss_bwd(const float* s, const float* t, const float* cosv, float* ds, int B, int P, float coef,
                                    int accumulate, void* stream) {
  DCLIP_REQUIRE(s && t && cosv && ds && B > 0 && P > 0 && P % 4 == 0, "cosine_loss_bwd: bad arguments");
  hipLaunchKernelGGL(cosine_bwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, s, t, cosv, ds, B, P, 1e-12f,
                     coef, accumulate);
  DCLIP_CHECK_LAUNCH("cosine_loss_bwd");
  return DCLIP_OK;
}

DCLIP_API int dclip_sub_reduce(const float* a, const float* b, float* out, int n, float scale, int accumulate,
                               void* stream) {
  DCLIP_REQUIRE(a && out && n > 0, "sub_reduce: bad arguments");
  hipLaunchKernelGGL(sub_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, b, out, n, scale, 0.0f, accumulate);
  DCLIP_CHECK_LAUNCH("sub_reduce");
  return DCLIP_OK;
}
