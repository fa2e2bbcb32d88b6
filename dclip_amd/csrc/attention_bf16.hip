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
#include <stdlib.h>

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

// ---------------------------------------------------------------------------------------------------------------
// Whole-head form for the sequence lengths of the CLIP towers (S <= 288: 50, 77, 197, 257 tokens): one workgroup per
// (batch, head), one wave per 32 queries (NB = ceil(S / 32) waves), the head's whole K and V staged in LDS ONCE — the
// tiled kernel above re-reads them once per 64-query tile and pads 257 tokens to 320 on both axes.  With every key
// on chip the softmax is exact in two passes instead of online: pass 1 forms S^T = K Q^T block by block and keeps only
// the row maximum (a lane owns one query: 16 v_max per 32-key block, one cross-half shuffle at the end); pass 2 forms
// the scores again, p = 2^(c s - c max) is one FMA + one v_exp_f32 per score, and P V accumulates with no rescaling of
// the accumulator and no running statistics — the online form's VALU work (rescale, alpha, masked selects on every
// block) was twice its MFMA work at head dim 64.  Only the block that holds the sequence end (and, causal, the
// diagonal block) runs the masked variant.
// XQ (S == 32 NB + 1, not causal: the 257 tokens of ViT-L/14): NB waves own the first 32 NB queries and SHARE the last one —
// a ninth wave for one query makes the workgroup 9 waves, of which only one fits a CU at this register count (16 wave slots
// at 4 per SIMD); with 8 waves two fit.  Wave w forms the last query's scores against key block w (wave 0 also the block
// that holds the last key) with that query in every column of the B operand, its local maximum / sum / P V go to LDS, and
// after one barrier wave 0 merges the NB + 1 partial results (the flash-attention merge, once per head).
template <int NB, bool CAUSAL, bool XQ>
__global__ void __launch_bounds__(64 * NB, (XQ ? 4 : 1)) attn_fwd_bf16_head_kernel(const unsigned short* __restrict__ qkv,
                                                                     unsigned short* __restrict__ out, int S, int H,
                                                                     float* __restrict__ lse = nullptr) {
  constexpr int KB = NB + (XQ ? 1 : 0);            // 32-key blocks staged
  constexpr int CHUNKS = KB * 256, NTHR = 64 * NB, ITER = (CHUNKS + NTHR - 1) / NTHR;
  __shared__ __attribute__((aligned(16))) unsigned char Ks[KB * 32 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char Vs[KB * 32 * 128];
  __shared__ __attribute__((aligned(16))) float xpart[XQ ? KB * 68 : 4];      // per key block: max, sum, pad, pad, O[64]
  __shared__ __attribute__((aligned(16))) unsigned short xq_row[XQ ? 64 : 8];  // the shared query
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int D = H * HD, ld = 3 * D;
  const int query = wave * 32 + l31;
  const unsigned short* base = qkv + (size_t)b * S * ld + h * HD;

  // Q fragments (B operand): Q[query][16 s + 8 half .. +7] — requested FIRST, in front of the K / V staging loads, so
  // that they do not become a second round trip to memory behind the barrier
  bf16x8 qf[4];
  {
    const unsigned short* qrow = base + (size_t)min(query, S - 1) * ld + 8 * half;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qrow + 16 * s));
  }
  __builtin_amdgcn_sched_barrier(0);
  // all eight 16-byte loads of a thread go out together (rows past the end read the last row and are zeroed after: a
  // load under a per-row condition is waited for on its own — four serial round trips to memory per workgroup)
  {
    u32x4 kv[ITER], vv[ITER], xqv = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int c = 0; c < ITER; ++c) {
      const int id = tid + c * NTHR, rc = min(id >> 3, S - 1), g = id & 7;
      kv[c] = *reinterpret_cast<const u32x4*>(base + (size_t)rc * ld + D + g * 8);
      vv[c] = *reinterpret_cast<const u32x4*>(base + (size_t)rc * ld + 2 * D + g * 8);
    }
    if (XQ && tid < 8) xqv = *reinterpret_cast<const u32x4*>(base + (size_t)(S - 1) * ld + tid * 8);
#pragma unroll
    for (int c = 0; c < ITER; ++c) {
      const int id = tid + c * NTHR, row = id >> 3, g = id & 7;
      if (row >= S) kv[c] = vv[c] = u32x4{0u, 0u, 0u, 0u};
      if (CHUNKS % NTHR == 0 || id < CHUNKS) {
        *reinterpret_cast<u32x4*>(Ks + gran_off(row, g)) = kv[c];
        *reinterpret_cast<u32x4*>(Vs + gran_off(row, g)) = vv[c];
      }
    }
    if (XQ && tid < 8) *reinterpret_cast<u32x4*>(xq_row + tid * 8) = xqv;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) asm volatile("" ::"v"(qf[s]));   // keep the Q loads in front of the barrier (the optimizer sinks them)
  __syncthreads();
  if (wave * 32 >= S) return;                      // (never with NB = ceil(S / 32); keeps a mis-sized launch harmless)

  const int last = (S - 1) >> 5;                   // block that holds the last key
  const int nkb = CAUSAL ? min(wave, last) + 1 : last + 1;

  // K fragments of one 32-key block (A operand of S^T = K Q^T): 4 x ds_read_b128 into the SAME registers every block —
  // issued right after the block's score MFMAs have consumed the previous contents, so they land behind the softmax
  // VALU work and the P V MFMAs instead of in front of the next score chain.
  bf16x8 kf[4];
  auto load_k = [&](int kb) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
      kf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Ks + gran_off(32 * kb + l31, 2 * s + half)));
  };
  auto qk_with = [&](const bf16x8 (&q)[4]) {
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], q[s], st, 0, 0, 0);
    return st;
  };
  auto qk = [&]() { return qk_with(qf); };
  auto valid = [&](int kb, int r) {
    const int key = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * half;
    return key < S && (!CAUSAL || key <= query);
  };
  // The last block a wave visits is the only one that can hold masked keys (sequence end; causal: the diagonal): it is
  // peeled off, the loops over the full blocks before it have no branches.
  const int nfull = nkb - 1;

  // ---- pass 1: row maximum of the raw scores
  float mx = -INFINITY;
  load_k(0);
  for (int kb = 0; kb < nfull; ++kb) {
    const f32x16 st = qk();
    __builtin_amdgcn_sched_barrier(0);
    load_k(kb + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[r]);
  }
  {
    const f32x16 st = qk();
    __builtin_amdgcn_sched_barrier(0);
    load_k(0);                                       // pass 2 starts over
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, valid(nfull, r) ? st[r] : -INFINITY);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  constexpr float c = kScale * 1.44269504088896341f;   // scale x log2(e)
  const float nmc = -mx * c;

  // ---- pass 2: p = 2^(c s - c max), row sum, O^T += V^T P^T
  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float lsum = 0.f;
  const int i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
  // MFMA step t contracts the keys of registers 8t..8t+7 = {32 kb + 16 t + 4 half + (0..3)} and {.. + 8 + (0..3)}; the
  // A operand V^T[d][those keys] comes out of the row-major V tile through the transposing read (issued with the next
  // block's K fragments, before the exponentials)
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 vf[2][2];
  auto load_v = [&](int kb) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int k0 = 32 * kb + 16 * t + 4 * half;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int dcol = 32 * dt + 16 * ((lane >> 4) & 1) + 4 * pp;   // first head dim of this lane's 8-byte piece
        const s16x4 lo = lds_tr16(Vs + gran_off(k0 + qq, dcol >> 3) + (dcol & 7) * 2);
        const s16x4 hi = lds_tr16(Vs + gran_off(k0 + 8 + qq, dcol >> 3) + (dcol & 7) * 2);
        vf[t][dt] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
  };
  auto pv = [&](const f32x16& st) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bf16x8 pf;
#pragma unroll
      for (int e = 0; e < 8; ++e) pf[e] = (__bf16)st[8 * t + e];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[t][dt]), pf, o[dt], 0, 0, 0);
    }
  };
  for (int kb = 0; kb < nfull; ++kb) {
    f32x16 st = qk();
    __builtin_amdgcn_sched_barrier(0);
    load_k(kb + 1);
    load_v(kb);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c, nmc));
      lsum += st[r];
    }
    pv(st);
  }
  {
    f32x16 st = qk();
    __builtin_amdgcn_sched_barrier(0);
    load_v(nfull);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = valid(nfull, r) ? __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c, nmc)) : 0.f;
      lsum += st[r];
    }
    pv(st);
  }
  lsum += __shfl_xor(lsum, 32);
  if (query < S) {
    const float inv = 1.0f / lsum;
    // log-sum-exp of the SCALED scores, for the backward of the training path: p = exp(scale (s - max)) sums to lsum
    if (lse && half == 0) lse[(size_t)bh * S + query] = mx * kScale + __logf(lsum);
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
  if (XQ) {
    // ---- the shared last query: this wave's key block(s), that query in every column of the B operand
    bf16x8 qx[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qx[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xq_row + 16 * s + 8 * half));
    const int nx = wave == 0 ? 2 : 1;
    for (int x = 0; x < nx; ++x) {
      const int xb = x == 0 ? wave : NB;             // wave 0 also takes the block with the last key
      load_k(xb);
      f32x16 st = qk_with(qx);
      load_v(xb);
      float m2 = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * xb + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (key >= S) st[r] = -INFINITY;
        m2 = fmaxf(m2, st[r]);
      }
      m2 = fmaxf(m2, __shfl_xor(m2, 32));            // every block holds at least one key: finite
      float l2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[r] = __builtin_amdgcn_exp2f((st[r] - m2) * c);   // -inf -> 0
        l2 += st[r];
      }
      l2 += __shfl_xor(l2, 32);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
      pv(st);
      if (l31 == 0) {                                // all 32 columns are the same query: column 0 writes
        if (half == 0) {
          xpart[xb * 68 + 0] = m2;
          xpart[xb * 68 + 1] = l2;
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) xpart[xb * 68 + 4 + 32 * dt + (r & 3) + 8 * (r >> 2) + 4 * half] = o[dt][r];
      }
    }
    __syncthreads();
    if (wave == 0) {                                 // merge: lane d owns head dim d
      float mm = -INFINITY;
#pragma unroll
      for (int xb = 0; xb < KB; ++xb) mm = fmaxf(mm, xpart[xb * 68]);
      float num = 0.f, den = 0.f;
#pragma unroll
      for (int xb = 0; xb < KB; ++xb) {
        const float w = __builtin_amdgcn_exp2f((xpart[xb * 68] - mm) * c);
        den += w * xpart[xb * 68 + 1];
        num += w * xpart[xb * 68 + 4 + lane];
      }
      out[((size_t)b * S + (S - 1)) * D + h * HD + lane] = bf16_bits(num / den);
    }
  }
}

template <int NB>
void launch_head(const unsigned short* qkv, unsigned short* out, int B, int S, int H, int causal, hipStream_t st,
                 float* lse = nullptr) {
  if (causal) hipLaunchKernelGGL((attn_fwd_bf16_head_kernel<NB, true, false>), dim3(B * H), dim3(64 * NB), 0, st, qkv, out, S, H, lse);
  else hipLaunchKernelGGL((attn_fwd_bf16_head_kernel<NB, false, false>), dim3(B * H), dim3(64 * NB), 0, st, qkv, out, S, H, lse);
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the whole-head kernel for SHORT sequences (S <= 64: the 50 tokens of ViT-B/32) on the bf16 MFMAs — the
// training student of configs c3 / c5.  One workgroup (two waves) per (batch, head); Q, K, V, dO of the head are staged
// in LDS once as bf16 tiles; scores, softmax and dS are fp32, P and dS are rounded to bf16 for the three products they feed,
// dq / dk / dv leave as bf16 (the A operand of the qkv projection's data- and weight-gradient GEMMs).
// There is no exchange between the waves after the staging barrier.  Wave w plays two roles:
//   KEYS   32w..32w+31: for each query block, S = Q K^T and dP = dO V^T as C[query][key] — a lane owns ONE key column and 16
//          queries in registers, which is the B operand of dV^T[d][key] = dO^T[d][q] P[q][key] and dK^T = Q^T dS when MFMA step
//          t contracts the queries of registers 8t..8t+7 (the contraction order of a product is free; the A operands dO^T, Q^T
//          come out of the row-major tiles through the transposing LDS read, as V^T does in the forward kernels);
//   QUERIES 32w..32w+31: for each key block, S^T = K Q^T and dP^T = V dO^T as C[key][query] — a lane owns one query column
//          (lse and delta are per-lane scalars) and its registers are the B operand of dQ^T[d][q] = K^T[d][key] dS^T[key][q].
// S and dP are formed twice (once per layout): 8 extra MFMAs per block pair against an LDS transpose of dS and a second
// barrier — at the bf16 rate the whole kernel is ~60 MFMAs per wave and is bound by its loads.
template <bool CAUSAL>
__global__ void __launch_bounds__(128) attn_bwd_bf16_kernel(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ out,
                                                            const unsigned short* __restrict__ dout, const float* __restrict__ lse,
                                                            unsigned short* __restrict__ dqkv, int S, int H) {
  __shared__ __attribute__((aligned(16))) unsigned char Qs[64 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char Ks[64 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char Vs[64 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char dOs[64 * 128];
  __shared__ float lse_s[64], dl_s[64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int D = H * HD, ld = 3 * D;
  const unsigned short* base = qkv + (size_t)b * S * ld + h * HD;
  const unsigned short* obase = out + (size_t)b * S * D + h * HD;
  const unsigned short* dobase = dout + (size_t)b * S * D + h * HD;

  // ---- staging: 4 tiles x 64 rows x 8 granules over 128 threads = 4 granules per thread and tile, plus the O granules that
  // pair with this thread's dO granules (delta = rowsum(O * dO)); every load is requested before the first is used (rows
  // past the end read the last row and are zeroed)
  {
    u32x4 qv[4], kv[4], vv[4], dv[4], ov[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int id = tid + c * 128, rc = min(id >> 3, S - 1), g = id & 7;
      qv[c] = *reinterpret_cast<const u32x4*>(base + (size_t)rc * ld + g * 8);
      kv[c] = *reinterpret_cast<const u32x4*>(base + (size_t)rc * ld + D + g * 8);
      vv[c] = *reinterpret_cast<const u32x4*>(base + (size_t)rc * ld + 2 * D + g * 8);
      dv[c] = *reinterpret_cast<const u32x4*>(dobase + (size_t)rc * D + g * 8);
      ov[c] = *reinterpret_cast<const u32x4*>(obase + (size_t)rc * D + g * 8);
    }
    if (tid < 64) lse_s[tid] = lse[(size_t)bh * S + min(tid, S - 1)];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int id = tid + c * 128, row = id >> 3, g = id & 7;
      // delta: 8 elements here, the other 7 granules of the row sit in the 7 neighbouring lanes
      float dl = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned int o2 = ov[c][e], d2 = dv[c][e];
        dl += __builtin_bit_cast(float, o2 << 16) * __builtin_bit_cast(float, d2 << 16);
        dl += __builtin_bit_cast(float, o2 & 0xffff0000u) * __builtin_bit_cast(float, d2 & 0xffff0000u);
      }
      dl += __shfl_xor(dl, 1);
      dl += __shfl_xor(dl, 2);
      dl += __shfl_xor(dl, 4);
      if (g == 0) dl_s[row] = row < S ? dl : 0.f;
      if (row >= S) qv[c] = kv[c] = vv[c] = dv[c] = u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(Qs + gran_off(row, g)) = qv[c];
      *reinterpret_cast<u32x4*>(Ks + gran_off(row, g)) = kv[c];
      *reinterpret_cast<u32x4*>(Vs + gran_off(row, g)) = vv[c];
      *reinterpret_cast<u32x4*>(dOs + gran_off(row, g)) = dv[c];
    }
  }
  __syncthreads();
  const int nblk = (S + 31) >> 5;                   // 32-row blocks of queries = of keys (1 or 2)
  if (wave >= nblk) return;                         // (no barrier below)

  constexpr float c = kScale * 1.44269504088896341f;   // scale x log2(e)
  constexpr float l2e = 1.44269504088896341f;
  const int i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  // row fragment (A or B operand of a K = 64 contraction over head dims): X[row][16 s + 8 half .. +7], s = 0..3
  auto row_frag = [&](const unsigned char* tile, int row, bf16x8 (&f)[4]) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) f[s4] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(tile + gran_off(row, 2 * s4 + half)));
  };
  auto mm4 = [&](const bf16x8 (&a)[4], const bf16x8 (&bq)[4]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s4], bq[s4], acc, 0, 0, 0);
    return acc;
  };
  // X^T fragments for a contraction over the 32 rows r0..r0+31 of a row-major tile: [t][dt] = X^T[32 dt + ..][the rows that MFMA
  // step t contracts = r0 + 16 t + 4 half + (0..3) and + 8 + (0..3)] (see load_v in the forward kernel)
  auto tr_frags = [&](const unsigned char* tile, int r0, s16x8 (&f)[2][2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int k0 = r0 + 16 * t + 4 * half;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int dcol = 32 * dt + 16 * ((lane >> 4) & 1) + 4 * pp;
        const s16x4 lo = lds_tr16(tile + gran_off(k0 + qq, dcol >> 3) + (dcol & 7) * 2);
        const s16x4 hi = lds_tr16(tile + gran_off(k0 + 8 + qq, dcol >> 3) + (dcol & 7) * 2);
        f[t][dt] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
  };
  auto acc_xt = [&](f32x16 (&o)[2], const s16x8 (&xt)[2][2], const f32x16& w) {     // o[dt] += X^T (A) x w (B, rounded to bf16)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bf16x8 wf;
#pragma unroll
      for (int e = 0; e < 8; ++e) wf[e] = (__bf16)w[8 * t + e];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xt[t][dt]), wf, o[dt], 0, 0, 0);
    }
  };
  auto store_rows = [&](const f32x16 (&o)[2], int row, int part) {   // o[dt][4 j + e] = X[row][32 dt + 8 j + 4 half + e]
    unsigned short* orow = dqkv + ((size_t)b * S + row) * ld + part * D + h * HD;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u16x4 v = {bf16_bits(o[dt][4 * j + 0]), bf16_bits(o[dt][4 * j + 1]), bf16_bits(o[dt][4 * j + 2]), bf16_bits(o[dt][4 * j + 3])};
        *reinterpret_cast<u16x4*>(orow + 32 * dt + 8 * j + 4 * half) = v;
      }
  };

  // ---- role 1: this wave's KEYS (rows 32 wave + l31 of K and V), every query block -> dK, dV
  {
    const int key = 32 * wave + l31;
    bf16x8 kf[4], vf[4];
    row_frag(Ks, key, kf);
    row_frag(Vs, key, vf);
    f32x16 dk[2], dvv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dk[dt][r] = dvv[dt][r] = 0.f;
    for (int qb = CAUSAL ? wave : 0; qb < nblk; ++qb) {    // causal: query blocks before the key block hold no pair key <= query
      bf16x8 qf[4], df[4];
      row_frag(Qs, 32 * qb + l31, qf);
      row_frag(dOs, 32 * qb + l31, df);
      f32x16 st = mm4(qf, kf);                             // S[query][key]
      f32x16 dp = mm4(df, vf);                             // dP[query][key]
      s16x8 dot[2][2], qt[2][2];
      tr_frags(dOs, 32 * qb, dot);
      tr_frags(Qs, 32 * qb, qt);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = 32 * qb + (r & 3) + 8 * (r >> 2) + 4 * half;
        const bool ok = q < S && key < S && (!CAUSAL || key <= q);
        const float p = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c, -lse_s[q] * l2e)) : 0.f;
        st[r] = p;
        dp[r] = p * (dp[r] - dl_s[q]) * kScale;            // dS
      }
      acc_xt(dvv, dot, st);                                // dV^T[d][key] += dO^T[d][q] P[q][key]
      acc_xt(dk, qt, dp);                                  // dK^T[d][key] += Q^T[d][q] dS[q][key]
    }
    if (key < S) {
      store_rows(dk, key, 1);
      store_rows(dvv, key, 2);
    }
  }
  // ---- role 2: this wave's QUERIES, every key block -> dQ
  {
    const int query = 32 * wave + l31;
    bf16x8 qf[4], df[4];
    row_frag(Qs, query, qf);
    row_frag(dOs, query, df);
    const float nl = -lse_s[query] * l2e, dl = dl_s[query];
    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    const int nkb = CAUSAL ? wave + 1 : nblk;
    for (int kb = 0; kb < nkb; ++kb) {
      bf16x8 kf[4], vf[4];
      row_frag(Ks, 32 * kb + l31, kf);
      row_frag(Vs, 32 * kb + l31, vf);
      f32x16 st = mm4(kf, qf);                             // S^T[key][query]
      f32x16 dp = mm4(vf, df);                             // dP^T[key][query]
      s16x8 kt[2][2];
      tr_frags(Ks, 32 * kb, kt);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * half;
        const bool ok = key < S && query < S && (!CAUSAL || key <= query);
        const float p = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c, nl)) : 0.f;
        dp[r] = p * (dp[r] - dl) * kScale;                 // dS^T
      }
      acc_xt(dq, kt, dp);                                  // dQ^T[d][query] += K^T[d][key] dS^T[key][query]
    }
    if (query < S) store_rows(dq, query, 0);
  }
}


// ------------------------------------------------------------------------------------------- one query row per (sequence, head)
// The LAST layer of a frozen tower is needed for one row only: the CLS row of every image (hf:modeling_clip.py:650) or the
// first-EOS row of every caption (:574-581; causal: keys 0..row).  One WAVE per (sequence, head): lane j scores keys
// j, j + 64, ... against the query (128-byte key rows, the query row broadcast), the softmax runs across the wave, then
// lane d accumulates head dimension d of sum_j p_j V[j] (128-byte coalesced V rows, p_j by a wave shuffle).  Bound by
// reading K and V once: 2 x 2 B x S x 64 per (sequence, head) — the generic fp32 kernel it replaces here padded the one
// query row to a 64-row tile (390 us at 2048 crops x 12 heads x 50 tokens, against ~60 us of bytes).
constexpr int ROW_MAXI = 8;        // keys per lane: S <= 512
__global__ void __launch_bounds__(256) attn_row_fwd_bf16_kernel(const unsigned short* __restrict__ qkv, const int* __restrict__ rows,
                                                                unsigned short* __restrict__ out, int B, int S, int H) {
  const int lane = threadIdx.x & 63;
  const int bh = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bh >= B * H) return;
  const int b = bh / H, h = bh % H;
  const int D = H * HD, ld = 3 * D;
  const int qrow = rows ? min(max(rows[b], 0), S - 1) : 0;
  const int nkeys = rows ? qrow + 1 : S;
  const unsigned short* base = qkv + (size_t)b * S * ld + h * HD;
  float q[64];
  {
    const unsigned short* qp = base + (size_t)qrow * ld;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const u32x4 w = *reinterpret_cast<const u32x4*>(qp + 8 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        q[8 * c + 2 * e] = __builtin_bit_cast(float, w[e] << 16);
        q[8 * c + 2 * e + 1] = __builtin_bit_cast(float, w[e] & 0xffff0000u);
      }
    }
  }
  float sc[ROW_MAXI];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < ROW_MAXI; ++i) {
    const int j = lane + 64 * i;
    sc[i] = -INFINITY;
    if (64 * i < nkeys) {                                     // wave-uniform
      const unsigned short* kp = base + D + (size_t)min(j, nkeys - 1) * ld;
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(kp + 8 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a0 += q[8 * c + 2 * e] * __builtin_bit_cast(float, w[e] << 16);
          a1 += q[8 * c + 2 * e + 1] * __builtin_bit_cast(float, w[e] & 0xffff0000u);
        }
      }
      if (j < nkeys) sc[i] = (a0 + a1) * kScale;
      mx = fmaxf(mx, sc[i]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < ROW_MAXI; ++i) {
    sc[i] = (64 * i < nkeys && lane + 64 * i < nkeys) ? __expf(sc[i] - mx) : 0.f;
    sum += sc[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.f / sum;
  // lane d: out[d] = sum_j p_j V[j][d]
  const unsigned short* vp = base + 2 * D + lane;
  float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
  for (int i = 0; i < ROW_MAXI; ++i) {
    if (64 * i >= nkeys) break;                               // wave-uniform
    const int n = min(64, nkeys - 64 * i);
    int jj = 0;
    for (; jj + 1 < n; jj += 2) {
      const float p0 = __shfl(sc[i], jj), p1 = __shfl(sc[i], jj + 1);
      const float v0 = __builtin_bit_cast(float, (unsigned int)vp[(size_t)(64 * i + jj) * ld] << 16);
      const float v1 = __builtin_bit_cast(float, (unsigned int)vp[(size_t)(64 * i + jj + 1) * ld] << 16);
      acc0 += p0 * v0;
      acc1 += p1 * v1;
    }
    if (jj < n) acc0 += __shfl(sc[i], jj) * __builtin_bit_cast(float, (unsigned int)vp[(size_t)(64 * i + jj) * ld] << 16);
  }
  out[(size_t)b * D + h * HD + lane] = bf16_bits((acc0 + acc1) * inv);
}

}  // namespace

// One attention output row per sequence, bf16 in / bf16 out (the LAST layer of a frozen bf16 tower): rows == NULL: query row 0
// against all S keys (the CLS row of a vision tower); rows [B] int32: query row rows[b] against keys 0..rows[b] (the first-EOS
// row of a causal text tower).  qkv [B*S][3*H*64] bf16, out [B][H*64] bf16.  S <= 512.
DCLIP_API int dclip_attention_row_fwd_bf16(const void* qkv, const int32_t* rows, void* out, int B, int S, int H, void* stream) {
  DCLIP_REQUIRE(qkv && out, "attention_row_fwd_bf16: null pointer");
  DCLIP_REQUIRE(B > 0 && S > 0 && S <= 64 * ROW_MAXI && H > 0, "attention_row_fwd_bf16: B=%d S=%d (<= 512) H=%d", B, S, H);
  DCLIP_REQUIRE((uintptr_t)qkv % 16 == 0, "attention_row_fwd_bf16: 16-byte alignment");
  hipLaunchKernelGGL(attn_row_fwd_bf16_kernel, dim3(cdiv(B * H, 4)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)qkv,
                     (const int*)rows, (unsigned short*)out, B, S, H);
  DCLIP_CHECK_LAUNCH("attention_row_fwd_bf16");
  return DCLIP_OK;
}

// Training forms (bf16 student, configs c3 / c5): the forward also leaves the log-sum-exp of the scaled scores (S <= 288,
// the whole-head kernel), the backward (S <= 64) returns dq | dk | dv as bf16 [B*S][3*H*64].
DCLIP_API int dclip_attention_fwd_bf16_lse(const void* qkv, void* out, float* lse, int B, int S, int H, int causal, void* stream) {
  DCLIP_REQUIRE(qkv && out && lse, "attention_fwd_bf16_lse: null pointer");
  DCLIP_REQUIRE(B > 0 && S > 0 && S <= 288 && S != 257 && H > 0, "attention_fwd_bf16_lse: B=%d S=%d (<= 288, not 257) H=%d", B, S, H);
  DCLIP_REQUIRE(((uintptr_t)qkv | (uintptr_t)out) % 16 == 0, "attention_fwd_bf16_lse: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  const unsigned short* q = (const unsigned short*)qkv;
  unsigned short* o = (unsigned short*)out;
  switch (cdiv(S, 32)) {
    case 1: launch_head<1>(q, o, B, S, H, causal, st, lse); break;
    case 2: launch_head<2>(q, o, B, S, H, causal, st, lse); break;
    case 3: launch_head<3>(q, o, B, S, H, causal, st, lse); break;
    case 4: launch_head<4>(q, o, B, S, H, causal, st, lse); break;
    case 5: launch_head<5>(q, o, B, S, H, causal, st, lse); break;
    case 6: launch_head<6>(q, o, B, S, H, causal, st, lse); break;
    case 7: launch_head<7>(q, o, B, S, H, causal, st, lse); break;
    case 8: launch_head<8>(q, o, B, S, H, causal, st, lse); break;
    default: launch_head<9>(q, o, B, S, H, causal, st, lse); break;
  }
  DCLIP_CHECK_LAUNCH("attention_fwd_bf16_lse");
  return DCLIP_OK;
}

DCLIP_API int dclip_attention_bwd_bf16(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B,
                                       int S, int H, int causal, void* stream) {
  DCLIP_REQUIRE(qkv && out && dout && lse && dqkv, "attention_bwd_bf16: null pointer");
  DCLIP_REQUIRE(B > 0 && S > 0 && S <= 64 && H > 0, "attention_bwd_bf16: B=%d S=%d (<= 64) H=%d", B, S, H);
  DCLIP_REQUIRE(((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dqkv) % 16 == 0, "attention_bwd_bf16: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  if (causal)
    hipLaunchKernelGGL((attn_bwd_bf16_kernel<true>), dim3(B * H), dim3(128), 0, st, (const unsigned short*)qkv,
                       (const unsigned short*)out, (const unsigned short*)dout, lse, (unsigned short*)dqkv, S, H);
  else
    hipLaunchKernelGGL((attn_bwd_bf16_kernel<false>), dim3(B * H), dim3(128), 0, st, (const unsigned short*)qkv,
                       (const unsigned short*)out, (const unsigned short*)dout, lse, (unsigned short*)dqkv, S, H);
  DCLIP_CHECK_LAUNCH("attention_bwd_bf16");
  return DCLIP_OK;
}

DCLIP_API int dclip_attention_fwd_bf16(const void* qkv, void* out, int B, int S, int H, int causal, void* stream) {
  DCLIP_REQUIRE(qkv && out, "attention_fwd_bf16: null pointer");
  DCLIP_REQUIRE(B > 0 && S > 0 && H > 0, "attention_fwd_bf16: bad shape B=%d S=%d H=%d", B, S, H);
  DCLIP_REQUIRE(((uintptr_t)qkv | (uintptr_t)out) % 16 == 0, "attention_fwd_bf16: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  static const bool tiled_only = getenv("DCLIP_ATTN16_TILED") && atoi(getenv("DCLIP_ATTN16_TILED")) != 0;   // A/B switch
  if (S <= 288 && !tiled_only) {                    // whole-head kernel: every CLIP tower (50 / 77 / 197 / 257 tokens)
    const unsigned short* q = (const unsigned short*)qkv;
    unsigned short* o = (unsigned short*)out;
    static const bool no_xq = getenv("DCLIP_ATTN16_NO_XQ") && atoi(getenv("DCLIP_ATTN16_NO_XQ")) != 0;   // A/B switch
    if (S == 257 && !causal && !no_xq) {            // 8 waves sharing the 257th query: two workgroups per CU instead of one
      hipLaunchKernelGGL((attn_fwd_bf16_head_kernel<8, false, true>), dim3(B * H), dim3(512), 0, st, q, o, S, H);
      DCLIP_CHECK_LAUNCH("attention_fwd_bf16.head_xq");
      return DCLIP_OK;
    }
    switch (cdiv(S, 32)) {
      case 1: launch_head<1>(q, o, B, S, H, causal, st); break;
      case 2: launch_head<2>(q, o, B, S, H, causal, st); break;
      case 3: launch_head<3>(q, o, B, S, H, causal, st); break;
      case 4: launch_head<4>(q, o, B, S, H, causal, st); break;
      case 5: launch_head<5>(q, o, B, S, H, causal, st); break;
      case 6: launch_head<6>(q, o, B, S, H, causal, st); break;
      case 7: launch_head<7>(q, o, B, S, H, causal, st); break;
      case 8: launch_head<8>(q, o, B, S, H, causal, st); break;
      default: launch_head<9>(q, o, B, S, H, causal, st); break;
    }
    DCLIP_CHECK_LAUNCH("attention_fwd_bf16.head");
    return DCLIP_OK;
  }
  dim3 grid(B * H, cdiv(S, 64)), block(128);
  if (causal) hipLaunchKernelGGL((attn_fwd_bf16_kernel<true>), grid, block, 0, st, (const unsigned short*)qkv, (unsigned short*)out, S, H);
  else hipLaunchKernelGGL((attn_fwd_bf16_kernel<false>), grid, block, 0, st, (const unsigned short*)qkv, (unsigned short*)out, S, H);
  DCLIP_CHECK_LAUNCH("attention_fwd_bf16");
  return DCLIP_OK;
}
