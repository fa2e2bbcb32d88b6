// Meta-teacher tail: cosine-to-mean softmax pooling (training/patch_text_aggregation.py:243-265) forward and
// backward, and the ragged packing of word-token embeddings (training/text_tokenizer.py:195-213 +
// training/patch_text_aggregation.py:606-620).  One workgroup per sample; x[b] is L x E <= 80 x 768 floats and
// stays L2-resident over the three sweeps.
#include "common.h"

namespace {

constexpr int MAXL = 96;
constexpr float kCosEps = 1e-8f;

// wave-per-row dot products: s[l] = <x_l, m>, n[l] = |x_l|^2
__device__ __forceinline__ void rows_dot(const float* __restrict__ xb, const float* m_s, float* dot_s, float* nrm_s, int L,
                                         int E) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int l = wave; l < L; l += 4) {
    const f32x4* xr = reinterpret_cast<const f32x4*>(xb + (size_t)l * E);
    float d = 0.f, n = 0.f;
    for (int i = lane; i < (E >> 2); i += 64) {
      f32x4 v = xr[i], mm = *reinterpret_cast<const f32x4*>(m_s + 4 * i);
      d += (v[0] * mm[0] + v[1] * mm[1]) + (v[2] * mm[2] + v[3] * mm[3]);
      n += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    d = wave_sum(d);
    n = wave_sum(n);
    if (lane == 0) {
      dot_s[l] = d;
      nrm_s[l] = sqrtf(n);
    }
  }
}

// out[b,:] (+)= out_scale * sum_l softmax_l(cos(x_l, mean x)/T) x_l ;  weights[b,l] saved for the backward
__global__ void __launch_bounds__(256) aggregation_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                              float* __restrict__ weights, int L, int E, float inv_temp,
                                                              float out_scale, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // m[E] | dot[MAXL] | nrm[MAXL] | w[MAXL] | red[8]
  float* m_s = sm;
  float* dot_s = sm + E;
  float* nrm_s = dot_s + MAXL;
  float* w_s = nrm_s + MAXL;
  float* red = w_s + MAXL;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xb = x + (size_t)b * L * E;
  // sweep 1: mean over the sequence (thread per column)
  float msq = 0.f;
  for (int e = tid; e < E; e += 256) {
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += xb[(size_t)l * E + e];
    s /= (float)L;
    m_s[e] = s;
    msq += s * s;
  }
  msq = wave_sum(msq);
  if ((tid & 63) == 0) red[tid >> 6] = msq;
  __syncthreads();
  const float nm = fmaxf(sqrtf((red[0] + red[1]) + (red[2] + red[3])), kCosEps);
  // sweep 2: cosine of every row to the mean
  rows_dot(xb, m_s, dot_s, nrm_s, L, E);
  __syncthreads();
  // softmax over the sequence (one wave)
  if (tid < 64) {
    float mx = -INFINITY;
    for (int l = tid; l < L; l += 64) {
      float s = dot_s[l] / (fmaxf(nrm_s[l], kCosEps) * nm) * inv_temp;
      w_s[l] = s;
      mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int l = tid; l < L; l += 64) {
      float e = __expf(w_s[l] - mx);
      w_s[l] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int l = tid; l < L; l += 64) {
      w_s[l] *= inv;
      if (weights) weights[(size_t)b * L + l] = w_s[l];
    }
  }
  __syncthreads();
  // sweep 3: weighted sum
  for (int e = tid; e < E; e += 256) {
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += w_s[l] * xb[(size_t)l * E + e];
    s *= out_scale;
    float* o = out + (size_t)b * E + e;
    *o = accumulate ? *o + s : s;
  }
}

// dx_l = os*w_l*dout + ds_l (m/(a_l c) - s_l x_l / a_l^2 [|x_l|>eps]) + (1/L) sum_k ds_k (x_k/(a_k c) - s_k m/c^2 [|m|>eps])
//   with s = cosine, a_l = max(|x_l|,eps), c = max(|m|,eps), ds_l = os * w_l (g_l - sum_k w_k g_k) / T, g_l = <dout, x_l>
__global__ void __launch_bounds__(256) aggregation_bwd_kernel(const float* __restrict__ x, const float* __restrict__ weights,
                                                              const float* __restrict__ dout, float* __restrict__ dx, int L,
                                                              int E, float inv_temp, float out_scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // m[E] | u[E] | dot | nrm | g | ds | red[8]
  float* m_s = sm;
  float* u_s = sm + E;  // shared part of the gradient that reaches every row through the mean
  float* dot_s = u_s + E;
  float* nrm_s = dot_s + MAXL;
  float* g_s = nrm_s + MAXL;
  float* ds_s = g_s + MAXL;
  float* red = ds_s + MAXL;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const float* xb = x + (size_t)b * L * E;
  const float* w = weights + (size_t)b * L;
  const float* dob = dout + (size_t)b * E;
  float msq = 0.f;
  for (int e = tid; e < E; e += 256) {
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += xb[(size_t)l * E + e];
    s /= (float)L;
    m_s[e] = s;
    msq += s * s;
  }
  msq = wave_sum(msq);
  if (lane == 0) red[wave] = msq;
  __syncthreads();
  const float nm_raw = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
  const float c = fmaxf(nm_raw, kCosEps);
  rows_dot(xb, m_s, dot_s, nrm_s, L, E);
  // g_l = <dout, x_l>
  for (int l = wave; l < L; l += 4) {
    const f32x4* xr = reinterpret_cast<const f32x4*>(xb + (size_t)l * E);
    float d = 0.f;
    for (int i = lane; i < (E >> 2); i += 64) {
      f32x4 v = xr[i], dd = reinterpret_cast<const f32x4*>(dob)[i];
      d += (v[0] * dd[0] + v[1] * dd[1]) + (v[2] * dd[2] + v[3] * dd[3]);
    }
    d = wave_sum(d);
    if (lane == 0) g_s[l] = d;
  }
  __syncthreads();
  if (tid < 64) {
    float wg = 0.f;
    for (int l = tid; l < L; l += 64) wg += w[l] * g_s[l];
    wg = wave_sum(wg);
    float sum_ds_s = 0.f;
    for (int l = tid; l < L; l += 64) {
      float d = out_scale * w[l] * (g_s[l] - wg) * inv_temp;
      ds_s[l] = d;
      const float cosv = dot_s[l] / (fmaxf(nrm_s[l], kCosEps) * c);
      sum_ds_s += d * cosv;
    }
    sum_ds_s = wave_sum(sum_ds_s);
    if (tid == 0) red[4] = sum_ds_s;
  }
  __syncthreads();
  const float sum_ds_s = red[4];
  // u[e] = (1/L) ( sum_k ds_k x_k[e] / (a_k c)  -  [|m|>eps] m[e] sum_k ds_k s_k / c^2 )
  for (int e = tid; e < E; e += 256) {
    float s = 0.f;
    for (int k = 0; k < L; ++k) s += ds_s[k] / (fmaxf(nrm_s[k], kCosEps) * c) * xb[(size_t)k * E + e];
    if (nm_raw > kCosEps) s -= m_s[e] * sum_ds_s / (c * c);
    u_s[e] = s / (float)L;
  }
  __syncthreads();
  for (int l = wave; l < L; l += 4) {
    const float a = fmaxf(nrm_s[l], kCosEps);
    const float cosv = dot_s[l] / (a * c);
    const float k1 = out_scale * w[l];
    const float k2 = ds_s[l] / (a * c);
    const float k3 = (nrm_s[l] > kCosEps) ? ds_s[l] * cosv / (a * a) : 0.f;
    for (int e = lane; e < E; e += 64)
      dx[((size_t)b * L + l) * E + e] = k1 * dob[e] + k2 * m_s[e] - k3 * xb[(size_t)l * E + e] + u_s[e];
  }
}

// text[b, i, :] = tokens[b, 1+i, :] for i < n_b = max(eos[b]-1, 0); zero beyond; a caption with no word tokens
// contributes its sentence embedding as its single row.
__global__ void __launch_bounds__(256) pack_tokens_kernel(const float* __restrict__ tokens, const float* __restrict__ sentence,
                                                          const int32_t* __restrict__ eos, float* __restrict__ out, int T,
                                                          int Tmax, int P4, size_t total4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % P4);
    const size_t bt = i / P4;
    const int t = (int)(bt % Tmax);
    const size_t b = bt / Tmax;
    const int n = max(eos[b] - 1, 0);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t < n) v = reinterpret_cast<const f32x4*>(tokens)[(b * T + 1 + t) * P4 + d];
    else if (n == 0 && t == 0) v = reinterpret_cast<const f32x4*>(sentence)[b * P4 + d];
    reinterpret_cast<f32x4*>(out)[i] = v;
  }
}

// x[b, r, :] = 0 for r >= count[b]
__global__ void __launch_bounds__(256) mask_rows_kernel(float* __restrict__ x, const int32_t* __restrict__ count, int R, int E4,
                                                        size_t total4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t br = i / E4;
    const int r = (int)(br % R);
    const size_t b = br / R;
    if (r >= count[b]) reinterpret_cast<f32x4*>(x)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// NaN / Inf guards of the reference's glue (training/patch_text_aggregation.py:497-499 per region embedding, :542 per
// caption, :649 for the whole batch): a GROUP of `rows` consecutive rows that holds any non-finite value is replaced
// by zeros.  mode 0: detect (flags[g] = 1 / 0) and zero; mode 1: zero the groups whose flag is already set (backward).
__global__ void __launch_bounds__(256) sanitize_groups_kernel(float* __restrict__ x, int32_t* __restrict__ flags, size_t n4,
                                                              int mode) {
  f32x4* g = reinterpret_cast<f32x4*>(x) + (size_t)blockIdx.x * n4;
  int bad;
  if (mode == 0) {
    int mine = 0;
    for (size_t i = threadIdx.x; i < n4; i += 256) {
      const f32x4 v = g[i];
      // finite <=> |v| <= FLT_MAX; NaN compares false
#pragma unroll
      for (int e = 0; e < 4; ++e) mine |= (int)!(fabsf(v[e]) <= 3.402823466e38f);
    }
    bad = __syncthreads_or(mine);
    if (threadIdx.x == 0) flags[blockIdx.x] = bad ? 1 : 0;
  } else {
    bad = flags[blockIdx.x];
  }
  if (bad)
    for (size_t i = threadIdx.x; i < n4; i += 256) g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

inline int grid_for(size_t work) {
  size_t b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

DCLIP_API int dclip_aggregation_fwd(const float* x, float* out, float* weights, int B, int L, int E, float temperature,
                                    float out_scale, int accumulate, void* stream) {
  DCLIP_REQUIRE(x && out, "aggregation_fwd: null pointer");
  DCLIP_REQUIRE(B > 0 && L > 0 && L <= MAXL && E > 0 && E % 4 == 0 && E <= 4096, "aggregation_fwd: bad shape B=%d L=%d E=%d", B,
                L, E);
  const size_t lds = (size_t)(E + 3 * MAXL + 8) * sizeof(float);
  hipLaunchKernelGGL(aggregation_fwd_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, x, out, weights, L, E,
                     1.0f / temperature, out_scale, accumulate);
  DCLIP_CHECK_LAUNCH("aggregation_fwd");
  return DCLIP_OK;
}

DCLIP_API int dclip_aggregation_bwd(const float* x, const float* weights, const float* dout, float* dx, int B, int L, int E,
                                    float temperature, float out_scale, void* stream) {
  DCLIP_REQUIRE(x && weights && dout && dx, "aggregation_bwd: null pointer");
  DCLIP_REQUIRE(B > 0 && L > 0 && L <= MAXL && E > 0 && E % 4 == 0 && E <= 4096, "aggregation_bwd: bad shape");
  const size_t lds = (size_t)(2 * E + 4 * MAXL + 8) * sizeof(float);
  hipLaunchKernelGGL(aggregation_bwd_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, x, weights, dout, dx, L, E,
                     1.0f / temperature, out_scale);
  DCLIP_CHECK_LAUNCH("aggregation_bwd");
  return DCLIP_OK;
}

DCLIP_API int dclip_pack_tokens(const float* tokens, const float* sentence, const int32_t* eos, float* out, int B, int T,
                                int Tmax, int P, void* stream) {
  DCLIP_REQUIRE(tokens && sentence && eos && out, "pack_tokens: null pointer");
  DCLIP_REQUIRE(B > 0 && T > 0 && Tmax > 0 && Tmax <= T && P % 4 == 0, "pack_tokens: bad shape");
  const size_t total4 = (size_t)B * Tmax * P / 4;
  hipLaunchKernelGGL(pack_tokens_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, tokens, sentence, eos, out,
                     T, Tmax, P / 4, total4);
  DCLIP_CHECK_LAUNCH("pack_tokens");
  return DCLIP_OK;
}

DCLIP_API int dclip_mask_rows(float* x, const int32_t* count, int B, int R, int E, void* stream) {
  DCLIP_REQUIRE(x && count && B > 0 && R > 0 && E % 4 == 0, "mask_rows: bad arguments");
  const size_t total4 = (size_t)B * R * E / 4;
  hipLaunchKernelGGL(mask_rows_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, count, R, E / 4, total4);
  DCLIP_CHECK_LAUNCH("mask_rows");
  return DCLIP_OK;
}

DCLIP_API int dclip_sanitize_groups(float* x, int32_t* flags, int groups, int rows, int E, int mode, void* stream) {
  DCLIP_REQUIRE(x && flags && groups > 0 && rows > 0 && E > 0 && E % 4 == 0 && (mode == 0 || mode == 1),
                "sanitize_groups: bad arguments");
  hipLaunchKernelGGL(sanitize_groups_kernel, dim3(groups), dim3(256), 0, (hipStream_t)stream, x, flags,
                     (size_t)rows * E / 4, mode);
  DCLIP_CHECK_LAUNCH("sanitize_groups");
  return DCLIP_OK;
}
