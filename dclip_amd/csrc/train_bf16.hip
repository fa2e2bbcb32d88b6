// Helpers of the bf16 TRAINING path (BASELINE configs c3 / c5 quote the step in bf16): the student's forward, dgrad and
// wgrad GEMMs run on v_mfma_f32_32x32x16_bf16 through dclip_gemm_bf16 (C = A W^T, both operands K-major), fp32 master
// weights, fp32 residual stream / LayerNorm / softmax / attention core.  What the weight-gradient product needs on top of
// the forward kernels is its operands with the TOKEN index contiguous:  dW[out,in] = dY^T X = (dY^T)[out,tok] (X^T)[in,tok]^T.
//   transpose_to_bf16 : x [rows][cols] (fp32 or bf16) -> x^T [cols][ld >= rows] bf16, zero padded to ld, and optionally the
//                       untransposed bf16 copy in the same pass (dgrad's A operand) — HBM-bound, 64x64 tiles through LDS.
//   rowsum_bf16       : out[r] = sum_c x[r][c] over a bf16 matrix (bias gradient from an already transposed dY^T).
#include "common.h"

namespace {

typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_bits(float x) {
  __bf16 b = (__bf16)x;  // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  return __builtin_bit_cast(unsigned short, b);
}

constexpr int TT = 64;        // tile edge
constexpr int TPAD = TT + 2;  // LDS row stride in 16-bit elements: 132 bytes -> a column walk hits 32 distinct banks

// grid (ceil(cols/64), ceil(rows/64)); 256 threads.  X_BF16: input element type.
template <bool X_BF16>
__global__ void __launch_bounds__(256) transpose_to_bf16_kernel(const void* __restrict__ xv, unsigned short* __restrict__ yT,
                                                                unsigned short* __restrict__ ycopy, int rows, int cols, int ldx,
                                                                int ldyT, int ldy) {
  __shared__ unsigned short tile[TT * TPAD];
  const int r0 = blockIdx.y * TT, c0 = blockIdx.x * TT;
  const int tid = threadIdx.x;
  // load: 16 threads cover one row of 64 elements (4 each), 16 rows per pass, 4 passes
  const int lr = tid >> 4, lc = (tid & 15) * 4;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int r = r0 + pass * 16 + lr, c = c0 + lc;
    u16x4 b = {0, 0, 0, 0};
    if (r < rows) {
      if (X_BF16) {
        const unsigned short* x = reinterpret_cast<const unsigned short*>(xv) + (size_t)r * ldx + c;
        if (c + 3 < cols) b = *reinterpret_cast<const u16x4*>(x);
        else
          for (int e = 0; e < 4; ++e)
            if (c + e < cols) b[e] = x[e];
      } else {
        const float* x = reinterpret_cast<const float*>(xv) + (size_t)r * ldx + c;
        if (c + 3 < cols) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(x);
          b = u16x4{bf16_bits(v[0]), bf16_bits(v[1]), bf16_bits(v[2]), bf16_bits(v[3])};
        } else {
          for (int e = 0; e < 4; ++e)
            if (c + e < cols) b[e] = bf16_bits(x[e]);
        }
      }
      if (ycopy && c < cols) {
        unsigned short* y = ycopy + (size_t)r * ldy + c;
        if (c + 3 < cols) *reinterpret_cast<u16x4*>(y) = b;
        else
          for (int e = 0; e < 4; ++e)
            if (c + e < cols) y[e] = b[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[(pass * 16 + lr) * TPAD + lc + e] = b[e];   // rows past `rows` are zeros
  }
  __syncthreads();
  // store: output row = input column; 8 threads cover one output row of 64 elements (8 each = 16 bytes), 32 rows per pass
  const int oc = tid >> 3, seg = (tid & 7) * 8;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int c = c0 + pass * 32 + oc;        // input column = output row
    const int r = r0 + seg;                   // input row = output column
    if (c < cols && r < ldyT) {
      u16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = tile[(seg + e) * TPAD + pass * 32 + oc];
      *reinterpret_cast<u16x8*>(yT + (size_t)c * ldyT + r) = o;     // ldyT % 8 == 0 and r % 8 == 0: whole chunk in range
    }
  }
}

// All GEMM weights of a training tower in ONE launch: every fp32 master W [rows][cols] -> its bf16 copy [rows][ld] (the
// forward's operand) and its bf16 transpose W^T [cols][ldT] (the data-gradient GEMM's operand), 64x64 tiles, one workgroup
// per tile, the tensor found from a table of records (as the multi-tensor optimizer does).  The masters change every
// optimizer step, so this runs once per step: 96 launches (48 casts + 48 transposes for ViT-B) become one.
struct WeightRef {
  const float* src;
  unsigned short* dst;    // may be null
  unsigned short* dstT;   // may be null
  int rows, cols, ld, ldT;
  int tile0, tiles_c;     // first tile index of this tensor; tiles per tile-row
};

__global__ void __launch_bounds__(256) mt_weights_bf16_kernel(const WeightRef* __restrict__ refs, int ntensors) {
  __shared__ unsigned short tile[TT * TPAD];
  int lo = 0, hi = ntensors - 1;                      // last record with tile0 <= blockIdx.x (uniform: scalar code)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (refs[mid].tile0 <= (int)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const WeightRef t = refs[lo];
  const int local = blockIdx.x - t.tile0;
  const int r0 = (local / t.tiles_c) * TT, c0 = (local % t.tiles_c) * TT;
  const int tid = threadIdx.x;
  const int lr = tid >> 4, lc = (tid & 15) * 4;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int r = r0 + pass * 16 + lr, c = c0 + lc;
    u16x4 b = {0, 0, 0, 0};
    if (r < t.rows) {
      const float* x = t.src + (size_t)r * t.cols + c;
      if (c + 3 < t.cols) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x);
        b = u16x4{bf16_bits(v[0]), bf16_bits(v[1]), bf16_bits(v[2]), bf16_bits(v[3])};
      } else {
        for (int e = 0; e < 4; ++e)
          if (c + e < t.cols) b[e] = bf16_bits(x[e]);
      }
      if (t.dst && c < t.ld) *reinterpret_cast<u16x4*>(t.dst + (size_t)r * t.ld + c) = b;   // ld % 4 == 0; zeros past cols
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[(pass * 16 + lr) * TPAD + lc + e] = b[e];
  }
  if (!t.dstT) return;                                 // uniform
  __syncthreads();
  const int oc = tid >> 3, seg = (tid & 7) * 8;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int c = c0 + pass * 32 + oc, r = r0 + seg;
    if (c < t.cols && r < t.ldT) {
      u16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = tile[(seg + e) * TPAD + pass * 32 + oc];
      *reinterpret_cast<u16x8*>(t.dstT + (size_t)c * t.ldT + r) = o;
    }
  }
}

// one wave per row
__global__ void __launch_bounds__(256) rowsum_bf16_kernel(const unsigned short* __restrict__ x, float* __restrict__ out, int R,
                                                          int n, int ld) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const unsigned short* xr = x + (size_t)row * ld;
  float s = 0.f;
  const int n8 = n & ~7;
  for (int i = lane * 8; i < n8; i += 512) {
    const u16x8 b = *reinterpret_cast<const u16x8*>(xr + i);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += __builtin_bit_cast(float, (unsigned int)b[e] << 16);
  }
  for (int i = n8 + lane; i < n; i += 64) s += __builtin_bit_cast(float, (unsigned int)xr[i] << 16);
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

}  // namespace

DCLIP_API int dclip_transpose_to_bf16(const void* x, int x_is_bf16, void* yT, void* y_copy, int rows, int cols, int ldx,
                                      int ldyT, int ldy, void* stream) {
  DCLIP_REQUIRE(x && yT && rows > 0 && cols > 0, "transpose_to_bf16: bad arguments");
  DCLIP_REQUIRE(ldx >= cols && ldx % 4 == 0, "transpose_to_bf16: ldx must be >= cols and a multiple of 4");
  DCLIP_REQUIRE(ldyT >= rows && ldyT % 8 == 0, "transpose_to_bf16: ldyT must be >= rows and a multiple of 8");
  DCLIP_REQUIRE(!y_copy || (ldy >= cols && ldy % 4 == 0), "transpose_to_bf16: ldy must be >= cols and a multiple of 4");
  DCLIP_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)yT % 16 == 0 && (uintptr_t)y_copy % 8 == 0, "transpose_to_bf16: alignment");
  // the padded tail columns rows..ldyT-1 of every output row are written as zeros by the last row-tile
  dim3 grid(cdiv(cols, TT), cdiv(ldyT, TT));
  hipStream_t st = (hipStream_t)stream;
  if (x_is_bf16)
    hipLaunchKernelGGL((transpose_to_bf16_kernel<true>), grid, dim3(256), 0, st, x, (unsigned short*)yT, (unsigned short*)y_copy,
                       rows, cols, ldx, ldyT, ldy);
  else
    hipLaunchKernelGGL((transpose_to_bf16_kernel<false>), grid, dim3(256), 0, st, x, (unsigned short*)yT,
                       (unsigned short*)y_copy, rows, cols, ldx, ldyT, ldy);
  DCLIP_CHECK_LAUNCH("transpose_to_bf16");
  return DCLIP_OK;
}

DCLIP_API int dclip_mt_weights_record_bytes(void) { return (int)sizeof(WeightRef); }

DCLIP_API int dclip_mt_weights_bf16(const void* refs, int ntensors, int total_tiles, void* stream) {
  DCLIP_REQUIRE(refs && ntensors > 0 && total_tiles > 0, "mt_weights_bf16: bad arguments");
  hipLaunchKernelGGL(mt_weights_bf16_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const WeightRef*)refs,
                     ntensors);
  DCLIP_CHECK_LAUNCH("mt_weights_bf16");
  return DCLIP_OK;
}

DCLIP_API int dclip_rowsum_bf16(const void* x, float* out, int R, int n, int ld, void* stream) {
  DCLIP_REQUIRE(x && out && R > 0 && n > 0 && ld >= n && ld % 8 == 0, "rowsum_bf16: bad arguments");
  DCLIP_REQUIRE((uintptr_t)x % 16 == 0, "rowsum_bf16: alignment");
  hipLaunchKernelGGL(rowsum_bf16_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, out, R,
                     n, ld);
  DCLIP_CHECK_LAUNCH("rowsum_bf16");
  return DCLIP_OK;
}
