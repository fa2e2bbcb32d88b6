// bf16 attention forward for the FROZEN towers (opt-in precision="bf16"; BASELINE configs c3 / c5): the teacher's
// region encoder and the frozen text tower, hf:modeling_clip.py:298-335 without gradients.  q, k, v arrive as bf16
// (the qkv projection writes them directly), scores and the softmax are fp32, P is rounded to bf16 for the P V
// product, the context leaves as bf16 — the next GEMM's A operand.
//
// One wave per 32 queries, two waves per workgroup, keys streamed in tiles of 64 through LDS, flash-style running
// max / sum.  Scores are formed TRANSPOSED (S^T = K Q^T, v_mfma_f32_32x32x16_bf16): a lane then owns ONE query
// column (softmax statistics and the rescale of the accumulator are per-lane scalars, the only cross-lane step is
// the max / sum across the two half-waves) and its 16 accumulator registers of a 32-key block are keys
// {(r&3) + 8(r>>2) + 4 half}.  The contraction order of P V is free, so MFMA step t contracts exactly the keys a
// lane already holds in registers r = 8t..8t+7 — P goes register -> bf16 -> B operand with no LDS round trip — and
// the matching A operand V^T[d][those keys] is read straight out of the row-major V tile with the gfx950 transposing
// LDS read (ds_read_b64_tr_b16: 4 keys x 16 head dims per 16 lanes).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int HD = 64;             // head dim
constexpr int KT = 64;             // keys per LDS tile
constexpr float kScale = 0.125f;   // 64^-0.5

__device__ __forceinline__ unsigned short bf16_bits(float x) {
  __bf16 b = (__bf16)x;
  return __builtin_bit_cast(unsigned short, b);
}

// byte offset of 16-byte granule g (0..7) of row `row` in a [rows][64] bf16 tile
__device__ __forceinline__ int gran_off(int row, int g) { return row * 128 + ((g ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ s16x4 lds_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

template <bool CAUSAL>
__global__ void __launch_bounds__(128) attn_fwd_bf16_kernel(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ out,
                                                            int S, int H) {
  __shared__ __attribute__((aligned(16))) unsigned char Ks[KT * 128];
  __shared__ __attribute__((aligned(16))) unsigned char Vs[KT * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int D = H * HD, ld = 3 * D;
  const int q0 = blockIdx.y * 64 + wave * 32;
  const int query = q0 + l31;
  const unsigned short* base = qkv + (size_t)b * S * ld + h * HD;

  // Q fragments (B operand): Q[query][16 s + 8 half .. +7]
  bf16x8 qf[4];
  {
    const unsigned short* qrow = base + (size_t)min(query, S - 1) * ld + 8 * half;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qrow + 16 * s));
  }
  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m = -INFINITY, l = 0.f;

  int nkt = (S + KT - 1) / KT;
  if (CAUSAL) nkt = min(nkt, (int)(blockIdx.y * 64 + 63) / KT + 1);   // workgroup-uniform
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt > 0) __syncthreads();
    // stage K and V tiles: 64 rows x 8 granules each, 128 threads -> 4 + 4 granules per thread
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int id = tid + c * 128, row = id >> 3, g = id & 7;
      const int key = kt * KT + row;
      u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
      if (key < S) {
        kv = *reinterpret_cast<const u32x4*>(base + (size_t)key * ld + D + g * 8);
        vv = *reinterpret_cast<const u32x4*>(base + (size_t)key * ld + 2 * D + g * 8);
      }
      *reinterpret_cast<u32x4*>(Ks + gran_off(row, g)) = kv;
      *reinterpret_cast<u32x4*>(Vs + gran_off(row, g)) = vv;
    }
    __syncthreads();
    const bool live = !CAUSAL || kt * KT <= q0 + 31;   // wave-uniform: this key tile holds keys <= some query of the wave
    if (!live) continue;

    // scores, transposed: st[sub][r] = S[query l31][key kt*64 + 32 sub + (r&3) + 8 (r>>2) + 4 half]
    f32x16 st[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[sub][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Ks + gran_off(32 * sub + l31, 2 * s + half)));
        st[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[sub], 0, 0, 0);
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * KT + 32 * sub + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = st[sub][r] * kScale;
        if (key >= S || (CAUSAL && key > query)) v = -INFINITY;
        st[sub][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);
    const float msafe = (mn == -INFINITY) ? 0.f : mn;
    const float alpha = __expf(m - msafe);   // m = -inf -> 0
    m = mn;
    float rs = 0.f;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = __expf(st[sub][r] - msafe);
        st[sub][r] = pr;
        rs += pr;
      }
    rs += __shfl_xor(rs, 32);
    l = l * alpha + rs;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;

    // O^T[d][query] += V^T[d][keys] P^T[keys][query], 16 keys per MFMA: step (sub, t) contracts the keys of
    // registers 8t..8t+7 = {32 sub + 16 t + 4 half + (0..3)} and {.. + 8 + (0..3)}
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[e] = (__bf16)st[sub][8 * t + e];
        const int k0 = 32 * sub + 16 * t + 4 * half;
        // transposing read: lane i = 4 q + p of each 16-lane group addresses row (key) k + q, head dims d0 + 4 p .. + 3,
        // and receives V[k .. k+3][d0 + i]
        const int i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int d0 = 32 * dt + 16 * ((lane >> 4) & 1);
          const int dcol = d0 + 4 * pp;   // first head dim of this lane's 8-byte piece
          const int ka = k0 + qq, kb = k0 + 8 + qq;
          const s16x4 lo = lds_tr16(Vs + gran_off(ka, dcol >> 3) + (dcol & 7) * 2);
          const s16x4 hi = lds_tr16(Vs + gran_off(kb, dcol >> 3) + (dcol & 7) * 2);
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, both), pf, o[dt], 0, 0, 0);
        }
      }
  }
  if (query < S) {
    const float inv = 1.0f / l;
    unsigned short* orow = out + ((size_t)b * S + query) * D + h * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u16x4 v = {bf16_bits(o[dt][4 * j + 0] * inv), bf16_bits(o[dt][4 * j + 1] * inv), bf16_bits(o[dt][4 * j + 2] * inv),
                   bf16_bits(o[dt][4 * j + 3] * inv)};
        *reinterpret_cast<u16x4*>(orow + 32 * dt + 8 * j + 4 * half) = v;
      }
  }
}

}  // namespace

DCLIP_API int dclip_attention_fwd_bf16(const void* qkv, void* out, int B, int S, int H, int causal, void* stream) {
  DCLIP_REQUIRE(qkv && out, "attention_fwd_bf16: null pointer");
  DCLIP_REQUIRE(B > 0 && S > 0 && H > 0, "attention_fwd_bf16: bad shape B=%d S=%d H=%d", B, S, H);
  DCLIP_REQUIRE(((uintptr_t)qkv | (uintptr_t)out) % 16 == 0, "attention_fwd_bf16: 16-byte alignment");
  dim3 grid(B * H, cdiv(S, 64)), block(128);
  hipStream_t st = (hipStream_t)stream;
  if (causal) hipLaunchKernelGGL((attn_fwd_bf16_kernel<true>), grid, block, 0, st, (const unsigned short*)qkv, (unsigned short*)out, S, H);
  else hipLaunchKernelGGL((attn_fwd_bf16_kernel<false>), grid, block, 0, st, (const unsigned short*)qkv, (unsigned short*)out, S, H);
  DCLIP_CHECK_LAUNCH("attention_fwd_bf16");
  return DCLIP_OK;
}
