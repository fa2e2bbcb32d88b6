// Optimiser tail of the step: global-norm gradient clipping (Trainer(gradient_clip_val=0.5),
// training/CLIP_image_distill_training.py:41) and AdamW (torch.optim.AdamW defaults,
// training/CLIP_image_distillation.py:680).  HBM-bound elementwise: 16 B/param read + 12 B/param written.
#include "common.h"

namespace {

// partial[blockIdx.x] = sum of squares of this block's grid-stride share (fixed order -> deterministic)
__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ x, size_t n, float* __restrict__ partial) {
  __shared__ float red[4];
  const size_t n4 = n >> 2;
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  if (blockIdx.x == 0)
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) s += x[i] * x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// coef = min(1, max_norm / (sqrt(sum partials) + 1e-6));  also returns the norm (torch.nn.utils.clip_grad_norm_)
__global__ void __launch_bounds__(256) clip_coef_kernel(const float* __restrict__ partial, int n, float max_norm,
                                                        float* __restrict__ coef, float* __restrict__ norm_out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float nrm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    if (norm_out) *norm_out = nrm;
    *coef = fminf(1.0f, max_norm / (nrm + 1e-6f));
  }
}

__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, size_t n, float lr, float beta1, float beta2,
                                                    float eps, float wd, float bc1, float bc2_sqrt,
                                                    const float* __restrict__ grad_scale) {
  const float gs = grad_scale ? *grad_scale : 1.0f;
  const float decay = 1.0f - lr * wd;
  const float step = lr / bc1;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    f32x4 gv = reinterpret_cast<const f32x4*>(g)[i] * gs;
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
    mv = mv * beta1 + gv * (1.0f - beta1);
    vv = vv * beta2 + gv * gv * (1.0f - beta2);
    f32x4 out;
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = pv[e] * decay - step * mv[e] / (sqrtf(vv[e]) / bc2_sqrt + eps);
    reinterpret_cast<f32x4*>(p)[i] = out;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0)
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
      const float gg = g[i] * gs;
      const float mm = m[i] * beta1 + gg * (1.0f - beta1);
      const float vv = v[i] * beta2 + gg * gg * (1.0f - beta2);
      p[i] = p[i] * decay - step * mm / (sqrtf(vv) / bc2_sqrt + eps);
      m[i] = mm;
      v[i] = vv;
    }
}

// ---- multi-tensor forms: one launch walks a device table of tensors, split into fixed-size chunks -------------
struct TensorRef {
  float* p;
  const float* g;
  float* m;
  float* v;
  unsigned long long n;
  int step;
  int chunk0;  // index of this tensor's first chunk in the launch
};
constexpr int MT_CHUNK = 32768;  // elements per workgroup

__device__ __forceinline__ int find_tensor(const TensorRef* __restrict__ refs, int ntensors, int chunk) {
  int lo = 0, hi = ntensors - 1;
  while (lo < hi) {  // last tensor whose chunk0 <= chunk
    const int mid = (lo + hi + 1) >> 1;
    if (refs[mid].chunk0 <= chunk) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

// partial[chunk] = sum of squares of the chunk
__global__ void __launch_bounds__(256) mt_sumsq_kernel(const TensorRef* __restrict__ refs, int ntensors,
                                                       float* __restrict__ partial) {
  __shared__ float red[4];
  const int chunk = blockIdx.x;
  const TensorRef t = refs[find_tensor(refs, ntensors, chunk)];
  const size_t beg = (size_t)(chunk - t.chunk0) * MT_CHUNK;
  const size_t end = beg + MT_CHUNK < t.n ? beg + MT_CHUNK : (size_t)t.n;
  const float* x = t.g;
  float s = 0.f;
  const bool vec = ((uintptr_t)x % 16) == 0;
  if (vec) {
    const size_t e4 = beg + ((end - beg) & ~(size_t)3);
    for (size_t i = beg + (size_t)threadIdx.x * 4; i < e4; i += 1024) {
      f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
      s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    for (size_t i = e4 + threadIdx.x; i < end; i += 256) s += x[i] * x[i];
  } else {
    for (size_t i = beg + threadIdx.x; i < end; i += 256) s += x[i] * x[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[chunk] = (red[0] + red[1]) + (red[2] + red[3]);
}

// COUPLED = false: AdamW (decoupled decay, torch.optim.AdamW);  COUPLED = true: Adam with L2 folded into the gradient
// (g += wd * p, torch.optim.Adam — the teacher trainer's optimizer, training/train_contrastive_teacher.py:245-248)
template <bool COUPLED>
__global__ void __launch_bounds__(256) mt_adamw_kernel(const TensorRef* __restrict__ refs, int ntensors, float lr, float beta1,
                                                       float beta2, float eps, float wd, const float* __restrict__ grad_scale) {
  const int chunk = blockIdx.x;
  const TensorRef t = refs[find_tensor(refs, ntensors, chunk)];
  const size_t beg = (size_t)(chunk - t.chunk0) * MT_CHUNK;
  const size_t end = beg + MT_CHUNK < t.n ? beg + MT_CHUNK : (size_t)t.n;
  const float gs = grad_scale ? *grad_scale : 1.0f;
  const float bc1 = 1.0f - powf(beta1, (float)t.step);
  const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)t.step));
  const float decay = COUPLED ? 1.0f : 1.0f - lr * wd, step = lr / bc1;
  const bool vec = (((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) % 16) == 0;
  size_t e4 = vec ? beg + ((end - beg) & ~(size_t)3) : beg;
  for (size_t i = beg + (size_t)threadIdx.x * 4; i < e4; i += 1024) {
    f32x4 pv = *reinterpret_cast<f32x4*>(t.p + i);
    f32x4 gv = *reinterpret_cast<const f32x4*>(t.g + i) * gs;
    if (COUPLED) {     // g + wd p as ONE fused multiply-add of the (already scaled) gradient, as torch's add(param, alpha=wd):
#pragma unroll          // where g ~ -wd p cancels, the other contraction a compiler may pick (fma(g, scale, wd p)) differs visibly
      for (int e = 0; e < 4; ++e) gv[e] = __builtin_fmaf(pv[e], wd, gv[e]);
    }
    f32x4 mv = *reinterpret_cast<f32x4*>(t.m + i);
    f32x4 vv = *reinterpret_cast<f32x4*>(t.v + i);
    mv = mv * beta1 + gv * (1.0f - beta1);
    vv = vv * beta2 + gv * gv * (1.0f - beta2);
    f32x4 out;
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = pv[e] * decay - step * mv[e] / (sqrtf(vv[e]) / bc2_sqrt + eps);
    *reinterpret_cast<f32x4*>(t.p + i) = out;
    *reinterpret_cast<f32x4*>(t.m + i) = mv;
    *reinterpret_cast<f32x4*>(t.v + i) = vv;
  }
  for (size_t i = e4 + threadIdx.x; i < end; i += 256) {
    const float gsc = t.g[i] * gs;            // the product first, then ONE fma: the same rounding as the vector path above
    const float gg = COUPLED ? __builtin_fmaf(t.p[i], wd, gsc) : gsc;
    const float mm = t.m[i] * beta1 + gg * (1.0f - beta1);
    const float vv = t.v[i] * beta2 + gg * gg * (1.0f - beta2);
    t.p[i] = t.p[i] * decay - step * mm / (sqrtf(vv) / bc2_sqrt + eps);
    t.m[i] = mm;
    t.v[i] = vv;
  }
}

inline int blocks_for(size_t n, int cap) {
  size_t b = (n / 4 + 255) / 256;
  return (int)(b < 1 ? 1 : (b > (size_t)cap ? cap : b));
}

}  // namespace

DCLIP_API int dclip_sumsq_blocks(size_t n) { return blocks_for(n, 256); }

DCLIP_API int dclip_sumsq_f32(const float* x, size_t n, float* partial, void* stream) {
  DCLIP_REQUIRE(x && partial && n > 0, "sumsq: bad arguments");
  DCLIP_REQUIRE((uintptr_t)x % 16 == 0, "sumsq: x must be 16-byte aligned");
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, n, partial);
  DCLIP_CHECK_LAUNCH("sumsq");
  return DCLIP_OK;
}

DCLIP_API int dclip_clip_coef(const float* partial, int n, float max_norm, float* coef, float* norm_out, void* stream) {
  DCLIP_REQUIRE(partial && coef && n > 0, "clip_coef: bad arguments");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n, max_norm, coef, norm_out);
  DCLIP_CHECK_LAUNCH("clip_coef");
  return DCLIP_OK;
}

DCLIP_API int dclip_adamw_f32(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int step, const float* grad_scale, void* stream) {
  DCLIP_REQUIRE(p && g && m && v && n > 0 && step > 0, "adamw: bad arguments");
  DCLIP_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, "adamw: 16-byte aligned tensors");
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                     beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale);
  DCLIP_CHECK_LAUNCH("adamw");
  return DCLIP_OK;
}

// ---- multi-tensor entry points.  `refs` is a DEVICE array of ntensors records
//   { float* p; const float* g; float* m; float* v; uint64 n; int32 step; int32 chunk0; }   (48 bytes, see
//   dclip_mt_record_bytes / dclip_mt_chunk_elems), chunk0 = running sum of ceil(n / dclip_mt_chunk_elems()).
DCLIP_API int dclip_mt_record_bytes(void) { return (int)sizeof(TensorRef); }
DCLIP_API int dclip_mt_chunk_elems(void) { return MT_CHUNK; }

DCLIP_API int dclip_mt_sumsq_f32(const void* refs, int ntensors, int total_chunks, float* partial, void* stream) {
  DCLIP_REQUIRE(refs && partial && ntensors > 0 && total_chunks > 0, "mt_sumsq: bad arguments");
  hipLaunchKernelGGL(mt_sumsq_kernel, dim3(total_chunks), dim3(256), 0, (hipStream_t)stream, (const TensorRef*)refs, ntensors,
                     partial);
  DCLIP_CHECK_LAUNCH("mt_sumsq");
  return DCLIP_OK;
}

DCLIP_API int dclip_mt_adamw_f32(const void* refs, int ntensors, int total_chunks, float lr, float beta1, float beta2,
                                 float eps, float weight_decay, const float* grad_scale, void* stream) {
  DCLIP_REQUIRE(refs && ntensors > 0 && total_chunks > 0, "mt_adamw: bad arguments");
  hipLaunchKernelGGL(mt_adamw_kernel<false>, dim3(total_chunks), dim3(256), 0, (hipStream_t)stream, (const TensorRef*)refs,
                     ntensors, lr, beta1, beta2, eps, weight_decay, grad_scale);
  DCLIP_CHECK_LAUNCH("mt_adamw");
  return DCLIP_OK;
}

DCLIP_API int dclip_mt_adam_f32(const void* refs, int ntensors, int total_chunks, float lr, float beta1, float beta2,
                                float eps, float weight_decay, const float* grad_scale, void* stream) {
  DCLIP_REQUIRE(refs && ntensors > 0 && total_chunks > 0, "mt_adam: bad arguments");
  hipLaunchKernelGGL(mt_adamw_kernel<true>, dim3(total_chunks), dim3(256), 0, (hipStream_t)stream, (const TensorRef*)refs,
                     ntensors, lr, beta1, beta2, eps, weight_decay, grad_scale);
  DCLIP_CHECK_LAUNCH("mt_adam");
  return DCLIP_OK;
}
