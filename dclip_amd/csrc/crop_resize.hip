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
constexpr int DCLIP_FILTER_BILINEAR = 0, DCLIP_FILTER_BICUBIC = 1;

__device__ __forceinline__ double bilinear_filter(double x) {
  if (x < 0.0) x = -x;
  if (x < 1.0) return 1.0 - x;
  return 0.0;
}

// Pillow's BICUBIC: Keys cubic with a = -0.5, support 2.
__device__ __forceinline__ double bicubic_filter(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

__device__ __forceinline__ double filter_at(int filter, double x) {
  return filter == DCLIP_FILTER_BICUBIC ? bicubic_filter(x) : bilinear_filter(x);
}

// plan[r] = (image index, x1, y1, x2, y2, outW, outH, left, top): the source box, the size of the FULL resized image
// and the S x S window of it that is produced (regions: outW = outH = S, window at 0,0).
constexpr int PLAN = 9;

__global__ void plan_from_boxes_kernel(const int32_t* __restrict__ boxes, int32_t* __restrict__ plan, int NR, int S) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= NR) return;
  for (int i = 0; i < 5; ++i) plan[r * PLAN + i] = boxes[r * 5 + i];
  plan[r * PLAN + 5] = S;
  plan[r * PLAN + 6] = S;
  plan[r * PLAN + 7] = 0;
  plan[r * PLAN + 8] = 0;
}

// CLIPImageProcessor geometry: shortest edge -> S (the long edge = int(S * long / short), float64 division then
// truncation as in Python), then the centred S x S window: top = (newH - S) / 2, left = (newW - S) / 2.
__global__ void plan_shortest_edge_kernel(const int32_t* __restrict__ dims, int32_t* __restrict__ plan, int B, int S) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int h = dims[2 * b], w = dims[2 * b + 1];
  const int shrt = w <= h ? w : h, lng = w <= h ? h : w;
  const int new_long = (int)((double)((long long)S * lng) / (double)shrt);
  const int newW = w <= h ? S : new_long, newH = w <= h ? new_long : S;
  int32_t* p = plan + b * PLAN;
  p[0] = b; p[1] = 0; p[2] = 0; p[3] = w; p[4] = h;
  p[5] = newW; p[6] = newH; p[7] = (newW - S) / 2; p[8] = (newH - S) / 2;
}

// One thread per (region, axis, output index).
// bounds [NR][2][S][2] = (first input index, count); kk [NR][2][S][KS] fixed-point weights.
__global__ void __launch_bounds__(256) resize_coeffs_kernel(const int32_t* __restrict__ plan, int NR, int S, int KS,
                                                            int filter, int32_t* __restrict__ bounds,
                                                            int32_t* __restrict__ kk) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= NR * 2 * S) return;
  const int xx = t % S, axis = (t / S) % 2, r = t / (2 * S);
  const int32_t* bx = plan + (size_t)r * PLAN;
  const int inSize = axis == 0 ? bx[3] - bx[1] : bx[4] - bx[2];
  const int outSize = bx[5 + axis], xo = xx + bx[7 + axis];
  double scale, filterscale;
  scale = filterscale = (double)inSize / outSize;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = (filter == DCLIP_FILTER_BICUBIC ? 2.0 : 1.0) * filterscale;
  const double center = 0.0 + (xo + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > inSize) xmax = inSize;
  xmax -= xmin;
  if (xmax < 0) xmax = 0;
  if (xmax > KS) xmax = KS;  // cannot happen when KS = ceil(support)*2+1 of the largest source extent
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += filter_at(filter, (x + xmin - center + 0.5) * ss);
  int32_t* k = kk + (size_t)t * KS;
  for (int x = 0; x < KS; ++x) {
    int v = 0;
    if (x < xmax) {
      double w = filter_at(filter, (x + xmin - center + 0.5) * ss);
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
                                                       const int32_t* __restrict__ plan, const int32_t* __restrict__ bounds,
                                                       const int32_t* __restrict__ kk, uint8_t* __restrict__ tmp, int NR, int S,
                                                       int KS, int Hmax, int Wmax, int Hc) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)NR * Hc * S) return;
  const int xx = (int)(t % S), y = (int)((t / S) % Hc), r = (int)(t / ((size_t)S * Hc));
  const int32_t* bx = plan + (size_t)r * PLAN;
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

struct Norm {
  int on;
  float mean[3], stdv[3];
};

// ToTensor(): uint8 / 255 in fp32.  CLIPImageProcessor: float32(float64(v) * (1/255)), then (x - mean) / std in fp32.
__device__ __forceinline__ float finish(int v, const Norm& nm, int c) {
  if (!nm.on) return (float)v / 255.0f;
  const float x = (float)((double)v * 0.00392156862745098);
  return (x - nm.mean[c]) / nm.stdv[c];
}

// vertical pass + output transform: out[r][c][yy][xx] = finish(clip8(sum_y tmp[r][ymin+y][xx][c] * k[y]))
__global__ void __launch_bounds__(256) resize_v_kernel(const uint8_t* __restrict__ tmp, const int32_t* __restrict__ bounds,
                                                       const int32_t* __restrict__ kk, float* __restrict__ out, int NR, int S,
                                                       int KS, int Hc, Norm nm) {
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
  o[0] = finish(clip8(s0), nm, 0);
  o[plane] = finish(clip8(s1), nm, 1);
  o[2 * plane] = finish(clip8(s2), nm, 2);
}

inline int ksize_for(int max_in, int S, int filter) {
  double fs = (double)max_in / S;
  if (fs < 1.0) fs = 1.0;
  const double support = (filter == DCLIP_FILTER_BICUBIC ? 2.0 : 1.0) * fs;
  int c = (int)support;
  if ((double)c < support) ++c;  // ceil
  return c * 2 + 1;
}

struct Layout {
  size_t plan, bounds, kk, tmp, total;
  int KS;
};

inline Layout layout_for(int NR, int S, int max_h, int max_w, int filter) {
  Layout L;
  L.KS = ksize_for(max_h > max_w ? max_h : max_w, S, filter);
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  L.plan = 0;
  L.bounds = up((size_t)NR * PLAN * sizeof(int32_t));
  L.kk = L.bounds + up((size_t)NR * 2 * S * 2 * sizeof(int32_t));
  L.tmp = L.kk + up((size_t)NR * 2 * S * L.KS * sizeof(int32_t));
  L.total = L.tmp + (size_t)NR * max_h * S * 3;
  return L;
}

int run_resize(const uint8_t* images, const int32_t* dims, float* out, int Hmax, int Wmax, int NR, int S, int max_h,
               const Layout& L, int filter, const Norm& nm, void* workspace, hipStream_t st) {
  char* ws = (char*)workspace;
  const int32_t* plan = (const int32_t*)(ws + L.plan);
  int32_t* bounds = (int32_t*)(ws + L.bounds);
  int32_t* kk = (int32_t*)(ws + L.kk);
  uint8_t* tmp = (uint8_t*)(ws + L.tmp);
  hipLaunchKernelGGL(resize_coeffs_kernel, dim3(cdiv(NR * 2 * S, 256)), dim3(256), 0, st, plan, NR, S, L.KS, filter, bounds,
                     kk);
  DCLIP_CHECK_LAUNCH("crop_resize.coeffs");
  const size_t nh = (size_t)NR * max_h * S;
  hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)cdivz(nh, 256)), dim3(256), 0, st, images, dims, plan, bounds, kk, tmp, NR,
                     S, L.KS, Hmax, Wmax, max_h);
  DCLIP_CHECK_LAUNCH("crop_resize.h");
  const size_t nv = (size_t)NR * S * S;
  hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)cdivz(nv, 256)), dim3(256), 0, st, tmp, bounds, kk, out, NR, S, L.KS,
                     max_h, nm);
  DCLIP_CHECK_LAUNCH("crop_resize.v");
  return DCLIP_OK;
}

}  // namespace

DCLIP_API size_t dclip_crop_resize_workspace(int NR, int S, int max_crop_h, int max_crop_w) {
  return layout_for(NR, S, max_crop_h, max_crop_w, DCLIP_FILTER_BILINEAR).total;
}

DCLIP_API int dclip_crop_resize_u8(const uint8_t* images, const int32_t* dims, const int32_t* boxes, float* out, int B,
                                   int Hmax, int Wmax, int NR, int S, int max_crop_h, int max_crop_w, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  DCLIP_REQUIRE(images && dims && boxes && out, "crop_resize: null pointer");
  DCLIP_REQUIRE(B > 0 && Hmax > 0 && Wmax > 0 && NR > 0 && S > 0 && max_crop_h > 0 && max_crop_w > 0,
                "crop_resize: bad shape");
  const Layout L = layout_for(NR, S, max_crop_h, max_crop_w, DCLIP_FILTER_BILINEAR);
  if (!workspace || workspace_bytes < L.total) {
    dclip_set_error("crop_resize: workspace too small (%zu < %zu)", workspace_bytes, L.total);
    return DCLIP_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(plan_from_boxes_kernel, dim3(cdiv(NR, 256)), dim3(256), 0, st, boxes, (int32_t*)workspace, NR, S);
  DCLIP_CHECK_LAUNCH("crop_resize.plan");
  Norm nm{};
  return run_resize(images, dims, out, Hmax, Wmax, NR, S, max_crop_h, L, DCLIP_FILTER_BILINEAR, nm, workspace, st);
}

DCLIP_API size_t dclip_clip_preprocess_workspace(int B, int Hmax, int Wmax, int S) {
  return layout_for(B, S, Hmax, Wmax, DCLIP_FILTER_BICUBIC).total;
}

DCLIP_API int dclip_clip_preprocess_u8(const uint8_t* images, const int32_t* dims, float* out, int B, int Hmax, int Wmax,
                                       int S, const float* mean, const float* stdv, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  DCLIP_REQUIRE(images && dims && out && mean && stdv, "clip_preprocess: null pointer");
  DCLIP_REQUIRE(B > 0 && Hmax > 0 && Wmax > 0 && S > 0, "clip_preprocess: bad shape");
  const Layout L = layout_for(B, S, Hmax, Wmax, DCLIP_FILTER_BICUBIC);
  if (!workspace || workspace_bytes < L.total) {
    dclip_set_error("clip_preprocess: workspace too small (%zu < %zu)", workspace_bytes, L.total);
    return DCLIP_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(plan_shortest_edge_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, dims, (int32_t*)workspace, B, S);
  DCLIP_CHECK_LAUNCH("clip_preprocess.plan");
  Norm nm{};
  nm.on = 1;
  for (int c = 0; c < 3; ++c) {
    nm.mean[c] = mean[c];   // HOST pointers: three floats each, read here
    nm.stdv[c] = stdv[c];
  }
  return run_resize(images, dims, out, Hmax, Wmax, B, S, Hmax, L, DCLIP_FILTER_BICUBIC, nm, workspace, st);
}
