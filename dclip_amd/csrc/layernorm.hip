// LayerNorm forward / backward and column sums (HBM-bound; one wave per row, 16-byte accesses).
//
// Algorithmic bytes: fwd reads x and writes y (8 B/element); bwd reads dy, x (+dresidual) and writes dx
// (12-16 B/element).  Row statistics are two-pass in registers (mean, then centred variance), like ATen.
#include "common.h"

namespace {

constexpr int MAXC = 8;  // float4 chunks per lane -> D <= 2048

// EXACT: D == 256 NC (every CLIP width: 512, 768, 1024) — the `i < d4` tests compile away.  Loads are written as one
// unconditional group per row (x, gamma, beta / dy, x, dresidual: a lane past the row end re-reads chunk 0 and its
// value is discarded): a load under a per-chunk condition is waited for on its own (vmcnt(0), which also waits for
// every store issued before it), which made a row 4-7 serial round trips to memory.
template <int NC, bool EXACT>
__global__ void __launch_bounds__(256) ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                     int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int d4 = D >> 2;
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * D);
  f32x4 v[NC], g[NC], bt[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = lane + 64 * c, ic = (EXACT || i < d4) ? i : 0;
    v[c] = xr[ic];
    g[c] = reinterpret_cast<const f32x4*>(gamma)[ic];
    bt[c] = reinterpret_cast<const f32x4*>(beta)[ic];
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (!EXACT && lane + 64 * c >= d4) v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (EXACT || lane + 64 * c < d4) {
      f32x4 d = v[c] - mu;
      q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
  f32x4* yr = reinterpret_cast<f32x4*>(y + (size_t)row * D);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = lane + 64 * c;
    if (EXACT || i < d4) yr[i] = (v[c] - mu) * rs * g[c] + bt[c];
  }
}

// Each wave walks rows with a grid stride; per-column dgamma/dbeta partials stay in registers, are summed
// over the block's 4 waves through LDS and written to partial[blockIdx][NS][D].
// bf16 training path (NS == 3 / dx16): the same pass also leaves what the NEXT two launches of the backward would
// otherwise re-read dx for — its bf16 copy (the A operand of the following data- and weight-gradient GEMMs: no cast
// launch) and its column sums (the bias gradient of the Linear whose output gradient dx is: no colsum launch).
template <int NC, bool EXACT>
__global__ void __launch_bounds__(256) ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ dres,
                                                     float* __restrict__ dx, float* __restrict__ partial, int rows,
                                                     int D, unsigned short* __restrict__ dx16, int NS) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][NS][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d4 = D >> 2;
  f32x4 g[NC], dg[NC], db[NC], dsum[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    int i = lane + 64 * c;
    g[c] = (EXACT || i < d4) ? reinterpret_cast<const f32x4*>(gamma)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    dg[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    dsum[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float invD = 1.0f / (float)D;
  const bool has_res = dres != nullptr;                       // wave-uniform
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * D);
    const f32x4* dyr = reinterpret_cast<const f32x4*>(dy + (size_t)row * D);
    const f32x4* rr = reinterpret_cast<const f32x4*>((has_res ? dres : dy) + (size_t)row * D);   // (a valid address either way)
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[NC], dyh[NC], rv[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = lane + 64 * c, ic = (EXACT || i < d4) ? i : 0;
      dyh[c] = dyr[ic];
      xh[c] = xr[ic];
      rv[c] = rr[ic];
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (EXACT || lane + 64 * c < d4) {
        const f32x4 dyv = dyh[c];
        xh[c] = (xh[c] - mu) * rs;
        dg[c] += dyv * xh[c];
        db[c] += dyv;
        dyh[c] = dyv * g[c];
        f32x4 t = dyh[c] * xh[c];
        s1 += (t[0] + t[1]) + (t[2] + t[3]);
        s2 += (dyh[c][0] + dyh[c][1]) + (dyh[c][2] + dyh[c][3]);
      } else {
        xh[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        dyh[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const float c1 = wave_sum(s1) * invD, c2 = wave_sum(s2) * invD;
    f32x4* dxr = reinterpret_cast<f32x4*>(dx + (size_t)row * D);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = lane + 64 * c;
      if (EXACT || i < d4) {
        f32x4 o = (dyh[c] - c2 - xh[c] * c1) * rs;
        if (has_res) o += rv[c];
        dxr[i] = o;
        if (NS == 3) dsum[c] += o;
        if (dx16) {
          typedef unsigned short u16x4_t __attribute__((ext_vector_type(4)));
          u16x4_t b;
#pragma unroll
          for (int e = 0; e < 4; ++e) b[e] = __builtin_bit_cast(unsigned short, (__bf16)o[e]);
          *reinterpret_cast<u16x4_t*>(dx16 + (size_t)row * D + i * 4) = b;
        }
      }
    }
  }
  if (partial) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      int i = lane + 64 * c;
      if (EXACT || i < d4) {
        reinterpret_cast<f32x4*>(red + (wave * NS + 0) * D)[i] = dg[c];
        reinterpret_cast<f32x4*>(red + (wave * NS + 1) * D)[i] = db[c];
        if (NS == 3) reinterpret_cast<f32x4*>(red + (wave * NS + 2) * D)[i] = dsum[c];
      }
    }
    __syncthreads();
    const int W = NS * D;
    for (int i = threadIdx.x; i < W; i += 256) {
      float s = red[i] + red[W + i] + red[2 * W + i] + red[3 * W + i];
      partial[(size_t)blockIdx.x * W + i] = s;
    }
  }
}

// out[n] (+)= sum_p partial[p][n]   (fixed order -> deterministic).  Block = 16 columns x 64 row groups: N/16
// workgroups (96 for the 2 x 768 LayerNorm partials) instead of N/64, each thread adds P/64 values — the partials
// are L2 / Infinity-Cache resident (just written), so this is latency, not bandwidth: width buys time.
constexpr int RP_COLS = 16, RP_GROUPS = 64;
__global__ void __launch_bounds__(1024) reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out0,
                                                               float* __restrict__ out1, int P, int N0, int N1,
                                                               int accumulate, float* __restrict__ out2 = nullptr, int N2 = 0) {
  __shared__ float red[RP_GROUPS][RP_COLS + 1];
  __shared__ float red2[4][RP_COLS];
  const int cl = threadIdx.x % RP_COLS, rg = threadIdx.x / RP_COLS;
  const int n = blockIdx.x * RP_COLS + cl;
  const int N = N0 + N1 + N2;
  float s = 0.f;
  if (n < N)
    for (int p = rg; p < P; p += RP_GROUPS) s += partial[(size_t)p * N + n];
  red[rg][cl] = s;
  __syncthreads();
  // 64 -> 4 partial sums per column by 64 threads, then the last 4 by 16: always the same association
  if (threadIdx.x < RP_COLS * 4) {
    const int c = threadIdx.x % RP_COLS, q = threadIdx.x / RP_COLS;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[q * 16 + k][c];
    red2[q][c] = t;
  }
  __syncthreads();
  if (threadIdx.x >= RP_COLS || n >= N) return;
  s = (red2[0][cl] + red2[1][cl]) + (red2[2][cl] + red2[3][cl]);
  float* base = (n < N0) ? out0 : (n < N0 + N1 ? out1 : out2);
  if (!base) return;
  float* o = base + ((n < N0) ? n : (n < N0 + N1 ? n - N0 : n - N0 - N1));
  *o = (accumulate && n < N0 + N1) ? *o + s : s;      // (the column sums of dx are always assigned)
}

// column sums: block = 64 float4-columns x 4 row lanes; grid.y splits the rows
// bf16 form (the fc1 bias gradient of the bf16 training path: dh exists only as bf16 [tokens][I])
__global__ void __launch_bounds__(256) colsum_bf16_kernel(const unsigned short* __restrict__ X, float* __restrict__ partial, int M,
                                                          int N, int ldx) {
  typedef unsigned short u16x4_t __attribute__((ext_vector_type(4)));
  __shared__ f32x4 red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c4 = blockIdx.x * 64 + cl;
  const int n4 = N >> 2;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c4 < n4) {
    const int rows_per = (M + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    for (int r = r0 + rl; r < r1; r += 4) {
      const u16x4_t b = *reinterpret_cast<const u16x4_t*>(X + (size_t)r * ldx + c4 * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += __builtin_bit_cast(float, (unsigned int)b[e] << 16);
    }
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c4 < n4) {
    f32x4 t = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    *reinterpret_cast<f32x4*>(partial + (size_t)blockIdx.y * N + c4 * 4) = t;
  }
}

__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ X, float* __restrict__ partial, int M,
                                                     int N, int ldx) {
  __shared__ f32x4 red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c4 = blockIdx.x * 64 + cl;
  const int n4 = N >> 2;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c4 < n4) {
    const int rows_per = (M + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    for (int r = r0 + rl; r < r1; r += 4) s += *reinterpret_cast<const f32x4*>(X + (size_t)r * ldx + c4 * 4);
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c4 < n4) {
    f32x4 t = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    *reinterpret_cast<f32x4*>(partial + (size_t)blockIdx.y * N + c4 * 4) = t;
  }
}

inline int ln_bwd_blocks(int rows) {
  int b = cdiv(rows, 4);
  return b > 1024 ? 1024 : b;  // 4 workgroups per CU keep enough rows in flight to stream at HBM rate
}
inline int colsum_splits(int M) {
  int s = cdiv(M, 128);
  return s > 64 ? 64 : (s < 1 ? 1 : s);
}

}  // namespace

DCLIP_API int dclip_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                  float* rstd, int rows, int D, float eps, void* stream) {
  DCLIP_REQUIRE(x && gamma && beta && y, "layernorm_fwd: null pointer");
  DCLIP_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 256 * MAXC, "layernorm_fwd: D=%d must be a multiple of 4, <= %d",
                D, 256 * MAXC);
  dim3 grid(cdiv(rows, 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int nc = cdiv(D / 4, 64);
#define LN_FWD(NC, EX) hipLaunchKernelGGL((ln_fwd_kernel<NC, EX>), grid, block, 0, st, x, gamma, beta, y, mean, rstd, rows, D, eps)
  if (D == 512) LN_FWD(2, true);            // the CLIP widths: no per-chunk bounds tests
  else if (D == 768) LN_FWD(3, true);
  else if (D == 1024) LN_FWD(4, true);
  else if (nc <= 1) LN_FWD(1, false);
  else if (nc == 2) LN_FWD(2, false);
  else if (nc == 3) LN_FWD(3, false);
  else if (nc == 4) LN_FWD(4, false);
  else LN_FWD(8, false);
#undef LN_FWD
  DCLIP_CHECK_LAUNCH("layernorm_fwd");
  return DCLIP_OK;
}

DCLIP_API size_t dclip_layernorm_bwd_workspace(int rows, int D) {
  return (size_t)ln_bwd_blocks(rows) * 3 * D * sizeof(float);
}

DCLIP_API int dclip_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                                  const float* rstd, const float* dresidual, float* dx, float* dgamma,
                                  float* dbeta, int rows, int D, int accumulate_param_grads, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  return dclip_layernorm_bwd_ex(dy, x, gamma, mean, rstd, dresidual, dx, nullptr, dgamma, dbeta, nullptr, rows, D,
                                accumulate_param_grads, workspace, workspace_bytes, stream);
}

DCLIP_API int dclip_layernorm_bwd_ex(const float* dy, const float* x, const float* gamma, const float* mean,
                                     const float* rstd, const float* dresidual, float* dx, void* dx_bf16, float* dgamma,
                                     float* dbeta, float* dx_colsum, int rows, int D, int accumulate_param_grads,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  DCLIP_REQUIRE(dy && x && gamma && mean && rstd && dx, "layernorm_bwd: null pointer");
  DCLIP_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 256 * MAXC, "layernorm_bwd: bad D=%d", D);
  DCLIP_REQUIRE(!dx_bf16 || (uintptr_t)dx_bf16 % 8 == 0, "layernorm_bwd: dx_bf16 must be 8-byte aligned");
  const bool want_params = dgamma || dbeta || dx_colsum;
  const int blocks = ln_bwd_blocks(rows);
  if (want_params && (!workspace || workspace_bytes < dclip_layernorm_bwd_workspace(rows, D))) {
    dclip_set_error("layernorm_bwd: workspace too small (%zu < %zu)", workspace_bytes,
                    dclip_layernorm_bwd_workspace(rows, D));
    return DCLIP_EWORKSPACE;
  }
  float* partial = want_params ? (float*)workspace : nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int ns = dx_colsum ? 3 : 2;
  const size_t lds = want_params ? (size_t)4 * ns * D * sizeof(float) : 0;
  const int nc = cdiv(D / 4, 64);
  unsigned short* dx16 = (unsigned short*)dx_bf16;
#define LN_BWD(NC, EX)                                                                                                      \
  hipLaunchKernelGGL((ln_bwd_kernel<NC, EX>), dim3(blocks), dim3(256), lds, st, dy, x, gamma, mean, rstd, dresidual, dx, partial, \
                     rows, D, dx16, ns)
  if (D == 512) LN_BWD(2, true);
  else if (D == 768) LN_BWD(3, true);
  else if (D == 1024) LN_BWD(4, true);
  else if (nc <= 1) LN_BWD(1, false);
  else if (nc == 2) LN_BWD(2, false);
  else if (nc == 3) LN_BWD(3, false);
  else if (nc == 4) LN_BWD(4, false);
  else LN_BWD(8, false);
#undef LN_BWD
  DCLIP_CHECK_LAUNCH("layernorm_bwd");
  if (want_params) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(ns * D, RP_COLS)), dim3(1024), 0, st, partial, dgamma, dbeta, blocks, D,
                       D, accumulate_param_grads, dx_colsum, dx_colsum ? D : 0);
    DCLIP_CHECK_LAUNCH("layernorm_bwd.reduce");
  }
  return DCLIP_OK;
}

DCLIP_API size_t dclip_colsum_f32_workspace(int M, int N) { return (size_t)colsum_splits(M) * N * sizeof(float); }

DCLIP_API int dclip_colsum_f32(const float* X, float* out, int M, int N, int ldx, int accumulate, void* workspace,
                               size_t workspace_bytes, void* stream) {
  DCLIP_REQUIRE(X && out, "colsum: null pointer");
  DCLIP_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && ldx % 4 == 0 && ldx >= N, "colsum: bad shape M=%d N=%d ldx=%d", M, N, ldx);
  const int splits = colsum_splits(M);
  if (!workspace || workspace_bytes < (size_t)splits * N * sizeof(float)) {
    dclip_set_error("colsum: workspace too small");
    return DCLIP_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N / 4, 64), splits), dim3(256), 0, st, X, (float*)workspace, M, N, ldx);
  DCLIP_CHECK_LAUNCH("colsum");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(N, RP_COLS)), dim3(1024), 0, st, (const float*)workspace, out,
                     (float*)nullptr, splits, N, 0, accumulate);
  DCLIP_CHECK_LAUNCH("colsum.reduce");
  return DCLIP_OK;
}

// out[n] = sum_m X[m][n] over a bf16 matrix [M][ldx] (same workspace size as dclip_colsum_f32_workspace(M, N))
DCLIP_API int dclip_colsum_bf16(const void* X, float* out, int M, int N, int ldx, int accumulate, void* workspace,
                                size_t workspace_bytes, void* stream) {
  DCLIP_REQUIRE(X && out, "colsum_bf16: null pointer");
  DCLIP_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && ldx % 4 == 0 && ldx >= N, "colsum_bf16: bad shape M=%d N=%d ldx=%d", M, N, ldx);
  DCLIP_REQUIRE((uintptr_t)X % 8 == 0, "colsum_bf16: alignment");
  const int splits = colsum_splits(M);
  if (!workspace || workspace_bytes < (size_t)splits * N * sizeof(float)) {
    dclip_set_error("colsum_bf16: workspace too small");
    return DCLIP_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3(cdiv(N / 4, 64), splits), dim3(256), 0, st, (const unsigned short*)X,
                     (float*)workspace, M, N, ldx);
  DCLIP_CHECK_LAUNCH("colsum_bf16");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(N, RP_COLS)), dim3(1024), 0, st, (const float*)workspace, out,
                     (float*)nullptr, splits, N, 0, accumulate);
  DCLIP_CHECK_LAUNCH("colsum_bf16.reduce");
  return DCLIP_OK;
}
