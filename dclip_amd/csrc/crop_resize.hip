// Region-crop front end: box list -> crop -> Resize((S,S)) -> ToTensor(), on the GPU,
// BIT-EXACT with what the reference does on the host with PIL (training/image_tokenizer.py:28-32, :100-110):
// `image.crop(box)` (zero padding outside the image), Pillow's two-pass antialiased BILINEAR resample in 8-bit
// fixed point (horizontal pass to uint8, then vertical pass; 22-bit coefficients, round-half-up, clip to 0..255),
// then uint8 / 255 in fp32, CHW.  Integer/byte work: HBM-bound gathers, no matrix cores.
//
// The double-precision coefficient arithmetic follows Pillow's precompute_coeffs() operation by operation, with FMA
// contraction disabled so that it rounds exactly like the host code.
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int PREC = 32 - 8 - 2;  // Pillow PRECISION_BITS

__device__ __forceinline__ double bilinear_filter(double x) {
  if (x < 0.0) x = -x;
  if (x < 1.0) return 1.0 - x;
  return 0.0;
}

// boxes[r] = (image index, x1, y1, x2, y2).  One thread per (region, axis, output index).
// bounds [NR][2][S][2] = (first input index, count); kk [NR][2][S][KS] fixed-point weights.
__global__ void __launch_bounds__(256) resize_coeffs_kernel(const int32_t* __restrict__ boxes, int NR, int S, int KS,
                                                            int32_t* __restrict__ bounds, int32_t* __restrict__ kk) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= NR * 2 * S) return;
  const int xx = t % S, axis = (t / S) % 2, r = t / (2 * S);
  const int32_t* bx = boxes + (size_t)r * 5;
  const int inSize = axis == 0 ? bx[3] - bx[1] : bx[4] - bx[2];
  double scale, filterscale;
  scale = filterscale = (double)inSize / S;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;  // bilinear support = 1.0
  const double center = 0.0 + (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > inSize) xmax = inSize;
  xmax -= xmin;
  if (xmax < 0) xmax = 0;
  if (xmax > KS) xmax = KS;  // cannot happen when KS = ceil(support)*2+1 of the largest crop
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += bilinear_filter((x + xmin - center + 0.5) * ss);
  int32_t* k = kk + (size_t)t * KS;
  for (int x = 0; x < KS; ++x) {
    int v = 0;
    if (x < xmax) {
      double w = bilinear_filter((x + xmin - center + 0.5) * ss);
      if (ww != 0.0) w /= ww;
      v = (w < 0) ? (int)(-0.5 + w * (1 << PREC)) : (int)(0.5 + w * (1 << PREC));
    }
    k[x] = v;
  }
  bounds[(size_t)t * 2 + 0] = xmin;
  bounds[(size_t)t * 2 + 1] = xmax;
}

__device__ __forceinline__ int clip8(int v) {
  v >>= PREC;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: tmp[r][y][xx][c] for every row y of the crop
__global__ void __launch_bounds__(256) resize_h_kernel(const uint8_t* __restrict__ images, const int32_t* __restrict__ dims,
                                                       const int32_t* __restrict__ boxes, const int32_t* __restrict__ bounds,
                                                       const int32_t* __restrict__ kk, uint8_t* __restrict__ tmp, int NR, int S,
                                                       int KS, int Hmax, int Wmax, int Hc) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)NR * Hc * S) return;
  const int xx = (int)(t % S), y = (int)((t / S) % Hc), r = (int)(t / ((size_t)S * Hc));
  const int32_t* bx = boxes + (size_t)r * 5;
  const int b = bx[0], x1 = bx[1], y1 = bx[2], inH = bx[4] - bx[2];
  if (y >= inH) return;
  const int ih = dims[2 * b], iw = dims[2 * b + 1];
  const size_t ci = ((size_t)r * 2 + 0) * S + xx;
  const int xmin = bounds[ci * 2], xmax = bounds[ci * 2 + 1];
  const int32_t* k = kk + ci * KS;
  int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
  const int sy = y1 + y;
  if (sy >= 0 && sy < ih) {
    const uint8_t* row = images + ((size_t)b * Hmax + sy) * Wmax * 3;
    for (int x = 0; x < xmax; ++x) {
      const int sx = x1 + xmin + x;
      if (sx >= 0 && sx < iw) {  // image.crop() pads with zeros outside the image
        const int w = k[x];
        s0 += row[sx * 3 + 0] * w;
        s1 += row[sx * 3 + 1] * w;
        s2 += row[sx * 3 + 2] * w;
      }
    }
  }
  uint8_t* o = tmp + (((size_t)r * Hc + y) * S + xx) * 3;
  o[0] = (uint8_t)clip8(s0);
  o[1] = (uint8_t)clip8(s1);
  o[2] = (uint8_t)clip8(s2);
}

// vertical pass + ToTensor: out[r][c][yy][xx] = clip8(sum_y tmp[r][ymin+y][xx][c] * k[y]) / 255
__global__ void __launch_bounds__(256) resize_v_kernel(const uint8_t* __restrict__ tmp, const int32_t* __restrict__ bounds,
                                                       const int32_t* __restrict__ kk, float* __restrict__ out, int NR, int S,
                                                       int KS, int Hc) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)NR * S * S) return;
  const int xx = (int)(t % S), yy = (int)((t / S) % S), r = (int)(t / ((size_t)S * S));
  const size_t ci = ((size_t)r * 2 + 1) * S + yy;
  const int ymin = bounds[ci * 2], ymax = bounds[ci * 2 + 1];
  const int32_t* k = kk + ci * KS;
  int s0 = 1 << (PREC - 1), s1 = s0, s2 = s0;
  for (int y = 0; y < ymax; ++y) {
    const uint8_t* px = tmp + (((size_t)r * Hc + ymin + y) * S + xx) * 3;
    const int w = k[y];
    s0 += px[0] * w;
    s1 += px[1] * w;
    s2 += px[2] * w;
  }
  const size_t plane = (size_t)S * S;
  float* o = out + (size_t)r * 3 * plane + (size_t)yy * S + xx;
  o[0] = (float)clip8(s0) / 255.0f;
  o[plane] = (float)clip8(s1) / 255.0f;
  o[2 * plane] = (float)clip8(s2) / 255.0f;
}

inline int ksize_for(int max_in, int S) {
  double fs = (double)max_in / S;
  if (fs < 1.0) fs = 1.0;
  int c = (int)fs;
  if ((double)c < fs) ++c;  // ceil
  return c * 2 + 1;
}

}  // namespace

DCLIP_API size_t dclip_crop_resize_workspace(int NR, int S, int max_crop_h, int max_crop_w) {
  const int KS = ksize_for(max_crop_h > max_crop_w ? max_crop_h : max_crop_w, S);
  size_t coef = (size_t)NR * 2 * S * (KS + 2) * sizeof(int32_t);
  size_t tmp = (size_t)NR * max_crop_h * S * 3;
  return ((coef + 255) / 256) * 256 + tmp;
}

DCLIP_API int dclip_crop_resize_u8(const uint8_t* images, const int32_t* dims, const int32_t* boxes, float* out, int B,
                                   int Hmax, int Wmax, int NR, int S, int max_crop_h, int max_crop_w, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  DCLIP_REQUIRE(images && dims && boxes && out, "crop_resize: null pointer");
  DCLIP_REQUIRE(B > 0 && Hmax > 0 && Wmax > 0 && NR > 0 && S > 0 && max_crop_h > 0 && max_crop_w > 0,
                "crop_resize: bad shape");
  const size_t need = dclip_crop_resize_workspace(NR, S, max_crop_h, max_crop_w);
  if (!workspace || workspace_bytes < need) {
    dclip_set_error("crop_resize: workspace too small (%zu < %zu)", workspace_bytes, need);
    return DCLIP_EWORKSPACE;
  }
  const int KS = ksize_for(max_crop_h > max_crop_w ? max_crop_h : max_crop_w, S);
  int32_t* bounds = (int32_t*)workspace;
  int32_t* kk = bounds + (size_t)NR * 2 * S * 2;
  const size_t coef = (size_t)NR * 2 * S * (KS + 2) * sizeof(int32_t);
  uint8_t* tmp = (uint8_t*)workspace + ((coef + 255) / 256) * 256;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(resize_coeffs_kernel, dim3(cdiv(NR * 2 * S, 256)), dim3(256), 0, st, boxes, NR, S, KS, bounds, kk);
  DCLIP_CHECK_LAUNCH("crop_resize.coeffs");
  const size_t nh = (size_t)NR * max_crop_h * S;
  hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)cdivz(nh, 256)), dim3(256), 0, st, images, dims, boxes, bounds, kk, tmp, NR, S,
                     KS, Hmax, Wmax, max_crop_h);
  DCLIP_CHECK_LAUNCH("crop_resize.h");
  const size_t nv = (size_t)NR * S * S;
  hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)cdivz(nv, 256)), dim3(256), 0, st, tmp, bounds, kk, out, NR, S, KS,
                     max_crop_h);
  DCLIP_CHECK_LAUNCH("crop_resize.v");
  return DCLIP_OK;
}
