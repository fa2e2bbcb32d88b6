// Embedding plumbing around the towers (all HBM-bound gathers / scatters).
#include "common.h"

namespace {

inline int grid_for(size_t work) {
  size_t b = (work + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// cols[(b*g*g + gy*g + gx)][c*p*p + py*p + px] = pixels[b][c][gy*p+py][gx*p+px]
// One thread per destination element; consecutive threads walk px, so both sides move in runs of p floats.
__global__ void __launch_bounds__(256) im2col_kernel(const float* __restrict__ pix, float* __restrict__ cols, int B, int C,
                                                     int Himg, int Wimg, int p, int g, size_t total) {
  const int kdim = C * p * p;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % kdim);
    const size_t row = i / kdim;
    const int gx = (int)(row % g), gy = (int)((row / g) % g), b = (int)(row / ((size_t)g * g));
    const int px = k % p, py = (k / p) % p, c = k / (p * p);
    cols[i] = pix[(((size_t)b * C + c) * Himg + gy * p + py) * Wimg + gx * p + px];
  }
}

// Vector form for patch sizes that are multiples of 4 (ViT-B/32, ViT-B/16): a thread moves 4 consecutive px — 16 bytes
// in, 16 bytes (fp32) or 8 bytes (bf16, for the frozen towers' bf16 GEMM) out.  `ldc` = destination row length.
typedef unsigned short u16x4_e __attribute__((ext_vector_type(4)));
template <bool OUT_BF16>
__global__ void __launch_bounds__(256) im2col_vec_kernel(const float* __restrict__ pix, void* __restrict__ cols, int B, int C,
                                                         int Himg, int Wimg, int p, int g, int ldc, size_t total4) {
  const int kdim4 = C * p * p / 4, p4 = p / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const int k4 = (int)(i % kdim4);
    const size_t row = i / kdim4;
    const int gx = (int)(row % g), gy = (int)((row / g) % g), b = (int)(row / ((size_t)g * g));
    const int px = (k4 % p4) * 4, py = (k4 / p4) % p, c = k4 / (p4 * p);
    const f32x4 v = *reinterpret_cast<const f32x4*>(pix + (((size_t)b * C + c) * Himg + gy * p + py) * Wimg + gx * p + px);
    if (OUT_BF16) {
      u16x4_e o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        __bf16 h = (__bf16)v[e];
        o[e] = __builtin_bit_cast(unsigned short, h);
      }
      *reinterpret_cast<u16x4_e*>(reinterpret_cast<unsigned short*>(cols) + row * ldc + (size_t)k4 * 4) = o;
    } else {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(cols) + row * ldc + (size_t)k4 * 4) = v;
    }
  }
}

__global__ void __launch_bounds__(256) vision_assemble_fwd_kernel(const float* __restrict__ patch, const float* __restrict__ cls,
                                                                  const float* __restrict__ pos, float* __restrict__ x, int B,
                                                                  int S, int D4, size_t total4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D4);
    const size_t tok = i / D4;
    const int s = (int)(tok % S);
    const size_t b = tok / S;
    f32x4 v = (s == 0) ? reinterpret_cast<const f32x4*>(cls)[d]
                       : reinterpret_cast<const f32x4*>(patch)[(b * (S - 1) + (s - 1)) * D4 + d];
    reinterpret_cast<f32x4*>(x)[i] = v + reinterpret_cast<const f32x4*>(pos)[(size_t)s * D4 + d];
  }
}

// dpatch[b, i, :] = dx[b, 1+i, :]
__global__ void __launch_bounds__(256) vision_assemble_bwd_kernel(const float* __restrict__ dx, float* __restrict__ dpatch, int B,
                                                                  int S, int D4, size_t total4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D4);
    const size_t tok = i / D4;
    const int s = (int)(tok % (S - 1));
    const size_t b = tok / (S - 1);
    reinterpret_cast<f32x4*>(dpatch)[i] = reinterpret_cast<const f32x4*>(dx)[(b * S + s + 1) * D4 + d];
  }
}

__global__ void __launch_bounds__(256) text_embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ tok,
                                                             const float* __restrict__ pos, float* __restrict__ x, int T,
                                                             int D4, int vocab, size_t total4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D4);
    const size_t bt = i / D4;
    const int t = (int)(bt % T);
    int64_t id = ids[bt];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    reinterpret_cast<f32x4*>(x)[i] =
        reinterpret_cast<const f32x4*>(tok)[(size_t)id * D4 + d] + reinterpret_cast<const f32x4*>(pos)[(size_t)t * D4 + d];
  }
}

// dtok[ids[b,t], :] += dx[b,t,:]  (float atomics: rows repeat across the batch)
__global__ void __launch_bounds__(256) text_embed_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dx,
                                                             float* __restrict__ dtok, int D, int vocab, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const size_t bt = i / D;
    int64_t id = ids[bt];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    atomicAdd(dtok + (size_t)id * D + d, dx[i]);
  }
}

// one wave per caption: index of the first EOS id, 0 if none (argmax of an all-false row)
__global__ void __launch_bounds__(256) first_eos_kernel(const int64_t* __restrict__ ids, int32_t* __restrict__ idx, int B, int T,
                                                        int64_t eos) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  int best = 0x7fffffff;
  for (int t = lane; t < T; t += 64)
    if (ids[(size_t)b * T + t] == eos) best = min(best, t);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
  if (lane == 0) idx[b] = (best == 0x7fffffff) ? 0 : best;
}

__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx,
                                                          float* __restrict__ out, int S, int D4, size_t total4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D4);
    const size_t b = i / D4;
    const int s = idx ? idx[b] : 0;
    reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(x)[(b * S + s) * D4 + d];
  }
}

// dx[b,s,:] = (s == idx[b]) ? dout[b,:] : 0   — writes the whole [B,S,D] tensor in one pass
__global__ void __launch_bounds__(256) scatter_rows_kernel(const float* __restrict__ dout, const int32_t* __restrict__ idx,
                                                           float* __restrict__ dx, int S, int D4, size_t total4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D4);
    const size_t tok = i / D4;
    const int s = (int)(tok % S);
    const size_t b = tok / S;
    const int sel = idx ? idx[b] : 0;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (s == sel) v = reinterpret_cast<const f32x4*>(dout)[b * D4 + d];
    reinterpret_cast<f32x4*>(dx)[i] = v;
  }
}

}  // namespace

DCLIP_API int dclip_im2col(const float* pixels, float* cols, int B, int C, int Himg, int Wimg, int patch, void* stream) {
  DCLIP_REQUIRE(pixels && cols, "im2col: null pointer");
  DCLIP_REQUIRE(B > 0 && C > 0 && patch > 0 && Himg == Wimg && Himg % patch == 0, "im2col: bad shape %dx%d patch %d", Himg,
                Wimg, patch);
  const int g = Himg / patch;
  const size_t total = (size_t)B * g * g * C * patch * patch;
  if (patch % 4 == 0 && ((uintptr_t)pixels | (uintptr_t)cols) % 16 == 0)
    hipLaunchKernelGGL((im2col_vec_kernel<false>), dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream, pixels, cols, B,
                       C, Himg, Wimg, patch, g, C * patch * patch, total / 4);
  else
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, pixels, cols, B, C, Himg, Wimg,
                       patch, g, total);
  DCLIP_CHECK_LAUNCH("im2col");
  return DCLIP_OK;
}

DCLIP_API int dclip_im2col_bf16(const float* pixels, void* cols, int B, int C, int Himg, int Wimg, int patch, int ldc,
                                void* stream) {
  DCLIP_REQUIRE(pixels && cols, "im2col_bf16: null pointer");
  DCLIP_REQUIRE(B > 0 && C > 0 && patch > 0 && patch % 4 == 0 && Himg == Wimg && Himg % patch == 0,
                "im2col_bf16: bad shape %dx%d patch %d (patch must be a multiple of 4)", Himg, Wimg, patch);
  DCLIP_REQUIRE(ldc >= C * patch * patch && ldc % 4 == 0 && (uintptr_t)pixels % 16 == 0 && (uintptr_t)cols % 8 == 0,
                "im2col_bf16: ldc / alignment");
  const int g = Himg / patch;
  const size_t total4 = (size_t)B * g * g * C * patch * patch / 4;
  hipLaunchKernelGGL((im2col_vec_kernel<true>), dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, pixels, cols, B, C,
                     Himg, Wimg, patch, g, ldc, total4);
  DCLIP_CHECK_LAUNCH("im2col_bf16");
  return DCLIP_OK;
}

DCLIP_API int dclip_vision_assemble_fwd(const float* patch, const float* cls, const float* pos, float* x, int B, int S, int D,
                                        void* stream) {
  DCLIP_REQUIRE(patch && cls && pos && x, "vision_assemble_fwd: null pointer");
  DCLIP_REQUIRE(B > 0 && S > 1 && D % 4 == 0, "vision_assemble_fwd: bad shape");
  const size_t total4 = (size_t)B * S * D / 4;
  hipLaunchKernelGGL(vision_assemble_fwd_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, patch, cls, pos, x,
                     B, S, D / 4, total4);
  DCLIP_CHECK_LAUNCH("vision_assemble_fwd");
  return DCLIP_OK;
}

DCLIP_API int dclip_vision_assemble_bwd(const float* dx, float* dpatch, int B, int S, int D, void* stream) {
  DCLIP_REQUIRE(dx && dpatch, "vision_assemble_bwd: null pointer");
  DCLIP_REQUIRE(B > 0 && S > 1 && D % 4 == 0, "vision_assemble_bwd: bad shape");
  const size_t total4 = (size_t)B * (S - 1) * D / 4;
  hipLaunchKernelGGL(vision_assemble_bwd_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, dx, dpatch, B, S,
                     D / 4, total4);
  DCLIP_CHECK_LAUNCH("vision_assemble_bwd");
  return DCLIP_OK;
}

DCLIP_API int dclip_text_embed_fwd(const int64_t* ids, const float* tok, const float* pos, float* x, int B, int T, int D,
                                   int vocab, void* stream) {
  DCLIP_REQUIRE(ids && tok && pos && x, "text_embed_fwd: null pointer");
  DCLIP_REQUIRE(B > 0 && T > 0 && D % 4 == 0 && vocab > 0, "text_embed_fwd: bad shape");
  const size_t total4 = (size_t)B * T * D / 4;
  hipLaunchKernelGGL(text_embed_fwd_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, ids, tok, pos, x, T,
                     D / 4, vocab, total4);
  DCLIP_CHECK_LAUNCH("text_embed_fwd");
  return DCLIP_OK;
}

DCLIP_API int dclip_text_embed_bwd(const int64_t* ids, const float* dx, float* dtok, int B, int T, int D, int vocab,
                                   void* stream) {
  DCLIP_REQUIRE(ids && dx && dtok, "text_embed_bwd: null pointer");
  const size_t total = (size_t)B * T * D;
  hipLaunchKernelGGL(text_embed_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, ids, dx, dtok, D, vocab,
                     total);
  DCLIP_CHECK_LAUNCH("text_embed_bwd");
  return DCLIP_OK;
}

DCLIP_API int dclip_first_eos(const int64_t* ids, int32_t* idx, int B, int T, int64_t eos_id, void* stream) {
  DCLIP_REQUIRE(ids && idx && B > 0 && T > 0, "first_eos: bad arguments");
  hipLaunchKernelGGL(first_eos_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, ids, idx, B, T, eos_id);
  DCLIP_CHECK_LAUNCH("first_eos");
  return DCLIP_OK;
}

DCLIP_API int dclip_gather_rows(const float* x, const int32_t* idx, float* out, int B, int S, int D, void* stream) {
  DCLIP_REQUIRE(x && out && B > 0 && S > 0 && D % 4 == 0, "gather_rows: bad arguments");
  const size_t total4 = (size_t)B * D / 4;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, idx, out, S, D / 4,
                     total4);
  DCLIP_CHECK_LAUNCH("gather_rows");
  return DCLIP_OK;
}

DCLIP_API int dclip_scatter_rows(const float* dout, const int32_t* idx, float* dx, int B, int S, int D, void* stream) {
  DCLIP_REQUIRE(dout && dx && B > 0 && S > 0 && D % 4 == 0, "scatter_rows: bad arguments");
  const size_t total4 = (size_t)B * S * D / 4;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, dout, idx, dx, S, D / 4,
                     total4);
  DCLIP_CHECK_LAUNCH("scatter_rows");
  return DCLIP_OK;
}
