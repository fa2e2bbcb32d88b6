// Multi-head self-attention core (head_dim 64) on v_mfma_f32_16x16x4_f32, forward and backward.
//
// One workgroup (4 waves) owns a 64-row tile of one (batch, head); each wave owns 16 of those rows and
// sweeps 64-row tiles of the other index with an online softmax (fwd) or with the saved log-sum-exp (bwd),
// so the S x S probabilities never reach HBM.  CLIP sequences are short (50 / 77 / 197 / 257): whole
// problems sit in one or a few tiles and the kernel is bounded by reading q,k,v once (12 B/element/head).
//
// LDS image of a [64 rows][64 floats] tile: 256-B rows of sixteen 16-B slots, slot' = slot ^ (row & 15).
//   * as an MFMA A operand (or a B operand that is K-major, e.g. K in Q K^T) a lane (r = lane&15,
//     quarter = lane>>4) reads slot 4g+quarter of row r with one ds_read_b128 and gets the contraction
//     indices 16g + 4*quarter + {0..3}: four MFMAs per read, conflict free.
//   * as a B operand indexed [contraction row][column] (V in P V, K in dS K, ...) a lane reads
//     element (row = 16g + 4*quarter + r, col = 16*nt + (lane&15)) with ds_read_b32, conflict free.
// Probability-like tiles go from the accumulator layout (col on the lane) back to an A operand through a
// per-wave [16][68] LDS scratch.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int TS = 64;       // tile rows
constexpr int HD = 64;       // head dim
constexpr int SCR = 68;      // scratch row stride (floats)
constexpr float kScale = 0.125f;  // 64^-0.5

__device__ __forceinline__ int tile_off(int row, int col) {
  return row * HD + ((((col >> 2) ^ (row & 15)) << 2) | (col & 3));
}

// global rows [row0, row0+64) x 64 floats (row stride ld) -> swizzled LDS tile; rows >= nrows are zero.  In two steps so
// that the loads of SEVERAL tiles go out together: tile_fetch issues a thread's four 16-byte loads UNCONDITIONALLY (rows
// past the end read row nrows-1 and are zeroed in tile_commit) — a load under a per-row condition is waited for on its
// own, which made the staging of a workgroup 4 (per tile) serial round trips to memory.
__device__ __forceinline__ void tile_fetch(f32x4 (&v)[4], const float* __restrict__ g, int row0, int nrows, size_t ld) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int id = threadIdx.x + c * 256;
    const int row = min(row0 + (id >> 4), nrows - 1), slot = id & 15;
    v[c] = *reinterpret_cast<const f32x4*>(g + (size_t)row * ld + slot * 4);
  }
}
__device__ __forceinline__ void tile_commit(float* tile, f32x4 (&v)[4], int row0, int nrows) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int id = threadIdx.x + c * 256;
    const int row = id >> 4, slot = id & 15;
    if (row0 + row >= nrows) v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(tile + row * HD + ((slot ^ (row & 15)) << 2)) = v[c];
  }
}
__device__ __forceinline__ void stage_tile(float* tile, const float* __restrict__ g, int row0, int nrows, size_t ld) {
  f32x4 v[4];
  tile_fetch(v, g, row0, nrows, ld);
  tile_commit(tile, v, row0, nrows);
}
// two tiles over the same rows (K and V, Q and dO): eight loads in flight
__device__ __forceinline__ void stage_tiles2(float* ta, const float* __restrict__ ga, size_t lda, float* tb,
                                             const float* __restrict__ gb, size_t ldb, int row0, int nrows) {
  f32x4 va[4], vb[4];
  tile_fetch(va, ga, row0, nrows, lda);
  tile_fetch(vb, gb, row0, nrows, ldb);
  tile_commit(ta, va, row0, nrows);
  tile_commit(tb, vb, row0, nrows);
}

// fragment for rows [rbase, rbase+16): contraction group g
__device__ __forceinline__ f32x4 frag_k(const float* tile, int rbase, int g, int lane) {
  int row = rbase + (lane & 15);
  int slot = 4 * g + (lane >> 4);
  return *reinterpret_cast<const f32x4*>(tile + row * HD + ((slot ^ (row & 15)) << 2));
}

// acc[nt] (16 x 16 each) += A_rows(16 x 64 via regs af[g]) * T^T where T tile rows are the output columns
__device__ __forceinline__ void mma_rows_x_tileT(f32x4 (&acc)[4], const f32x4 (&af)[4], const float* tile, int lane) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 bf = frag_k(tile, 16 * nt, g, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[g][r], bf[r], acc[nt], 0, 0, 0);
    }
}

// acc[nt] (16 x 16 over columns 16nt..) += P(16 x 64, per-wave scratch) * T (64 x 64 tile, row = contraction)
__device__ __forceinline__ void mma_scratch_x_tile(f32x4 (&acc)[4], const float* scr, const float* tile, int lane) {
  const int qd = lane >> 4, l15 = lane & 15;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 pf = *reinterpret_cast<const f32x4*>(scr + l15 * SCR + 16 * g + 4 * qd);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float b = tile[tile_off(16 * g + 4 * qd + r, 16 * nt + l15)];
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf[r], b, acc[nt], 0, 0, 0);
      }
  }
}

// acc[nt] += A[rbase + (lane&15)][k] * B[k][16nt + (lane&15)], A by rows of a swizzled tile, B a swizzled tile
__device__ __forceinline__ void mma_tilerows_x_tile(f32x4 (&acc)[4], const float* atile, int rbase, const float* btile,
                                                    int lane) {
  const int qd = lane >> 4, l15 = lane & 15;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 af = frag_k(atile, rbase, g, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r], btile[tile_off(16 * g + 4 * qd + r, 16 * nt + l15)], acc[nt], 0,
                                                       0, 0);
  }
}

// acc[nt] += A[k][cbase + (lane&15)] * B[k][16nt + (lane&15)]   (A transposed on the fly: column reads)
__device__ __forceinline__ void mma_tilecols_x_tile(f32x4 (&acc)[4], const float* atile, int cbase, const float* btile,
                                                    int lane) {
  const int qd = lane >> 4, l15 = lane & 15;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float af[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) af[r] = atile[tile_off(16 * g + 4 * qd + r, cbase + l15)];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r], btile[tile_off(16 * g + 4 * qd + r, 16 * nt + l15)], acc[nt], 0,
                                                       0, 0);
  }
}

// ---- bf16 I/O of the short-sequence kernels (the bf16 TRAINING student, configs c3 / c5): q, k, v, the attention
// output, its gradient and dq / dk / dv travel as bf16 — half the bytes of kernels that are bound by them — while every
// product, the softmax and the accumulations stay in fp32 exactly as in the fp32 instances.
typedef unsigned short u16x4a __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8a __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 bf4_to_f32(u16x4a b) {
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = __builtin_bit_cast(float, (unsigned int)b[e] << 16);
  return v;
}
__device__ __forceinline__ u16x4a f32_to_bf4(f32x4 v) {
  u16x4a b;
#pragma unroll
  for (int e = 0; e < 4; ++e) b[e] = __builtin_bit_cast(unsigned short, (__bf16)v[e]);   // round to nearest even
  return b;
}
// 16 bytes = 8 consecutive bf16 of a row -> the two fp32 slots 2c, 2c+1 of the swizzled LDS image
__device__ __forceinline__ void commit8(float* tile, int row, int chunk, u16x8a b, bool zero) {
  f32x4 lo, hi;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    lo[e] = zero ? 0.f : __builtin_bit_cast(float, (unsigned int)b[e] << 16);
    hi[e] = zero ? 0.f : __builtin_bit_cast(float, (unsigned int)b[4 + e] << 16);
  }
  *reinterpret_cast<f32x4*>(tile + row * HD + (((2 * chunk) ^ (row & 15)) << 2)) = lo;
  *reinterpret_cast<f32x4*>(tile + row * HD + (((2 * chunk + 1) ^ (row & 15)) << 2)) = hi;
}
// a [64 rows][64] bf16 tile by 256 threads: 2 chunks of 8 per thread; requested unconditionally (see tile_fetch)
__device__ __forceinline__ void tile_fetch16(u16x8a (&v)[2], const unsigned short* __restrict__ g, int nrows, size_t ld) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int id = threadIdx.x + c * 256;
    v[c] = *reinterpret_cast<const u16x8a*>(g + (size_t)min(id >> 3, nrows - 1) * ld + (id & 7) * 8);
  }
}
__device__ __forceinline__ void tile_commit16(float* tile, u16x8a (&v)[2], int nrows) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int id = threadIdx.x + c * 256;
    commit8(tile, id >> 3, id & 7, v[c], (id >> 3) >= nrows);
  }
}

__device__ __forceinline__ void zero4(f32x4 (&a)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ------------------------------------------------------------------------------------------- forward
// q rows: q + (b*Sq + i)*ldq + h*64;  k/v rows: k + (b*Sk + j)*ldkv + h*64.  Self-attention passes the fused
// projection (k = q + D, v = q + 2D, ldq = ldkv = 3D, Sq = Sk); cross-attention passes separate buffers.
struct AttnArgs {
  const float* q;
  const float* k;
  const float* v;
  int ldq, ldkv;
  int Sq, Sk, H;
  const int32_t* q_rows;  // forward only, optional: ONE query row per batch at sequence position q_rows[b], attending
                          // keys <= q_rows[b] (the text tower's first-EOS row under the causal mask); then Sq == 1
};

template <bool CAUSAL>
__global__ void __launch_bounds__(256) attn_fwd_kernel(AttnArgs a, float* __restrict__ out, float* __restrict__ lse) {
  const int H = a.H, S = a.Sk, Sq = a.Sq;
  // Q, K, V tiles.  Once a wave holds its Q fragments in registers, its 16 rows of the Q tile become its private P
  // scratch (only that wave ever touches those rows), so P needs no extra LDS and no workgroup barrier.
  __shared__ __attribute__((aligned(16))) float lds[3 * TS * HD];
  float* Qs = lds;
  float* Ks = lds + TS * HD;
  float* Vs = lds + 2 * TS * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qd = lane >> 4, l15 = lane & 15;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int q0 = blockIdx.y * TS;
  const int D = H * HD;
  const int qpos = a.q_rows ? a.q_rows[b] : -1;          // >= 0: single query row at this sequence position
  const float* qbase = a.q + ((size_t)b * (qpos >= 0 ? S : Sq) + (qpos >= 0 ? qpos : 0)) * a.ldq + h * HD;
  const float* kbase = a.k + (size_t)b * S * a.ldkv + h * HD;
  const float* vbase = a.v + (size_t)b * S * a.ldkv + h * HD;

  {
    f32x4 vq[4], vk[4], vv[4];
    tile_fetch(vq, qbase, q0, Sq, (size_t)a.ldq);
    tile_fetch(vk, kbase, 0, S, (size_t)a.ldkv);
    tile_fetch(vv, vbase, 0, S, (size_t)a.ldkv);
    tile_commit(Qs, vq, q0, Sq);
    tile_commit(Ks, vk, 0, S);
    tile_commit(Vs, vv, 0, S);
  }
  __syncthreads();
  f32x4 qf[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) qf[g] = frag_k(Qs, 16 * wave, g, lane);

  float m[4], l[4];
  f32x4 o[4];
  zero4(o);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    m[r] = -INFINITY;
    l[r] = 0.f;
  }
  int nkt = (S + TS - 1) / TS;
  if (CAUSAL) nkt = min(nkt, (int)blockIdx.y + 1);
  if (qpos >= 0) nkt = min(nkt, qpos / TS + 1);
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt > 0) {
      __syncthreads();  // everyone is done with the previous K/V tile
      stage_tiles2(Ks, kbase, (size_t)a.ldkv, Vs, vbase, (size_t)a.ldkv, kt * TS, S);
      __syncthreads();
    }
    f32x4 s[4];
    zero4(s);
    mma_rows_x_tileT(s, qf, Ks, lane);
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int key = kt * TS + nt * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qrow = q0 + 16 * wave + 4 * qd + r;
        float v = s[nt][r] * kScale;
        if (key >= S || (CAUSAL && key > qrow) || (qpos >= 0 && key > qpos)) v = -INFINITY;
        s[nt][r] = v;
        mx[r] = fmaxf(mx[r], v);
      }
    }
    float alpha[4], rs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float mn = fmaxf(m[r], quarter_max(mx[r]));
      const float msafe = (mn == -INFINITY) ? 0.f : mn;
      alpha[r] = __expf(m[r] - msafe);  // m = -inf -> 0
      m[r] = mn;
      rs[r] = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        float p = __expf(s[nt][r] - msafe);
        s[nt][r] = p;
        rs[r] += p;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      l[r] = l[r] * alpha[r] + quarter_sum(rs[r]);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        o[nt][r] *= alpha[r];
        Qs[tile_off(16 * wave + 4 * qd + r, nt * 16 + l15)] = s[nt][r];   // this wave's private rows
      }
    }
    mma_tilerows_x_tile(o, Qs, 16 * wave, Vs, lane);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = q0 + 16 * wave + 4 * qd + r;
    if (row < Sq) {
      const float inv = 1.0f / l[r];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) out[((size_t)b * Sq + row) * D + h * HD + nt * 16 + l15] = o[nt][r] * inv;
      if (l15 == 0) lse[(size_t)bh * Sq + row] = m[r] + __logf(l[r]);
    }
  }
}

// ------------------------------------------------------------------------------------------- forward, whole rows
// Short self-attention (Sq == Sk <= 16*KT, KT <= 5: the 50-token ViT-B/32 and the 77-token text sequences): one
// workgroup per (batch, head), one wave per 16 query rows, the whole K and V of the head in LDS once.
//   * scores are produced TRANSPOSED (S^T = K Q^T): lane (l15, qd) then holds, for ITS query l15, the keys
//     16kt + 4qd + r — exactly the B operand P^T[key][query] of O^T = V^T P^T when MFMA step r contracts over the
//     keys {4qd + r}.  The contraction order of a dot product is free, so P never leaves the registers (no LDS
//     round trip, no online rescaling: the whole row is there), and O^T comes out with 4 consecutive head
//     dimensions per lane: 16-byte stores.
//   * 16-row granularity: a causal wave w touches key tiles 0..w only (77 tokens: 15 of the 25 16x16 tiles).
// IO16: a.q / a.k / a.v and `out` are bf16 (see the bf16 I/O note above); lse stays fp32.
template <int KT, bool CAUSAL, bool IO16 = false>
__global__ void __launch_bounds__(KT * 64) attn_fwd_rows_kernel(AttnArgs a, float* __restrict__ out,
                                                                float* __restrict__ lse) {
  constexpr int R = KT * 16;
  extern __shared__ __attribute__((aligned(16))) float lds_rows[];
  float* Ks = lds_rows;
  float* Vs = lds_rows + R * HD;
  const int H = a.H, S = a.Sk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qd = lane >> 4, l15 = lane & 15;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int D = H * HD;
  const int query = 16 * wave + l15;
  // Q never goes through LDS: a lane's B-operand fragments are 16-byte pieces of its own query row
  f32x4 qf[4];
  if constexpr (IO16) {
    const unsigned short* qrow =
        reinterpret_cast<const unsigned short*>(a.q) + ((size_t)b * S + min(query, S - 1)) * a.ldq + h * HD + 4 * qd;
#pragma unroll
    for (int g = 0; g < 4; ++g) qf[g] = bf4_to_f32(*reinterpret_cast<const u16x4a*>(qrow + 16 * g));
    // K and V: R * 8 chunks of 8 bf16 over KT * 64 threads = 2 per thread and tensor, all four requested together
    const unsigned short* src16[2] = {reinterpret_cast<const unsigned short*>(a.k) + (size_t)b * S * a.ldkv + h * HD,
                                      reinterpret_cast<const unsigned short*>(a.v) + (size_t)b * S * a.ldkv + h * HD};
    u16x8a stg[2][2];
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int id = threadIdx.x + c * (KT * 64);
        stg[w][c] = *reinterpret_cast<const u16x8a*>(src16[w] + (size_t)min(id >> 3, S - 1) * a.ldkv + (id & 7) * 8);
      }
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int id = threadIdx.x + c * (KT * 64);
        commit8(lds_rows + w * R * HD, id >> 3, id & 7, stg[w][c], (id >> 3) >= S);
      }
  } else {
    const float* qrow = a.q + ((size_t)b * S + min(query, S - 1)) * a.ldq + h * HD + 4 * qd;
#pragma unroll
    for (int g = 0; g < 4; ++g) qf[g] = *reinterpret_cast<const f32x4*>(qrow + 16 * g);   // rows >= S: never stored
  }
  // K and V of the head: R * 16 chunks over KT * 64 threads = 4 per thread and tensor, all eight loads requested before
  // the first is waited for (rows past the end read row S-1 and are zeroed: see tile_fetch)
  const float* src[2] = {a.k + (size_t)b * S * a.ldkv + h * HD, a.v + (size_t)b * S * a.ldkv + h * HD};
  if constexpr (!IO16) {
    f32x4 stg[2][4];
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int id = threadIdx.x + c * (KT * 64);
        stg[w][c] = *reinterpret_cast<const f32x4*>(src[w] + (size_t)min(id >> 4, S - 1) * a.ldkv + (id & 15) * 4);
      }
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int id = threadIdx.x + c * (KT * 64), row = id >> 4, slot = id & 15;
        if (row >= S) stg[w][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(lds_rows + w * R * HD + row * HD + ((slot ^ (row & 15)) << 2)) = stg[w][c];
      }
  }
  __syncthreads();
  const int nkt = CAUSAL ? wave + 1 : (S + 15) / 16;  // wave-uniform
  f32x4 st[KT];
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kt < nkt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 kf = frag_k(Ks, 16 * kt, g, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[r], qf[g][r], st[kt], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * kt + 4 * qd + r;
        float v = st[kt][r] * kScale;
        if (key >= S || (CAUSAL && key > query)) v = -INFINITY;
        st[kt][r] = v;
        mx = fmaxf(mx, v);
      }
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
    if (kt < nkt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(st[kt][r] - mx);
        st[kt][r] = p;
        sum += p;
      }
    }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  f32x4 o[4];
  zero4(o);
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
    if (kt < nkt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * kt + 4 * qd + r;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vs[tile_off(key, 16 * dt + l15)], st[kt][r], o[dt], 0, 0, 0);
      }
    }
  if (query < S) {
    const float inv = 1.0f / sum;
    if constexpr (IO16) {
      unsigned short* orow = reinterpret_cast<unsigned short*>(out) + ((size_t)b * S + query) * D + h * HD + 4 * qd;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<u16x4a*>(orow + 16 * dt) = f32_to_bf4(o[dt] * inv);
    } else {
      float* orow = out + ((size_t)b * S + query) * D + h * HD + 4 * qd;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(orow + 16 * dt) = o[dt] * inv;
    }
    if (qd == 0) lse[(size_t)bh * S + query] = mx + __logf(sum);
  }
}

// ------------------------------------------------------------------------------------------- forward, streamed
// Long self-attention (ViT-B/16's 197 tokens, ViT-L/14's 257): flash-style, same register discipline as the whole-row
// kernel but on v_mfma_f32_32x32x2_f32.  One wave per 32 queries, four waves per workgroup, keys streamed through
// 64-row LDS tiles.  Scores come out transposed (S^T = K Q^T): lane (l31, half) owns query column l31 and holds the
// keys {(r&3) + 8(r>>2) + 4 half} of a 32-key block in registers r = 0..15 — which is exactly the pair of keys one
// 32x32x2 MFMA step r of O^T = V^T P^T contracts (k-slot = half).  P never leaves registers; the running max / sum
// and the accumulator rescale are per-lane scalars; O^T leaves as 16-byte stores.
template <bool CAUSAL>
__global__ void __launch_bounds__(256) attn_fwd_stream_kernel(AttnArgs a, float* __restrict__ out, float* __restrict__ lse) {
  // (a double-buffered LDS-DMA staging of the K / V tiles was measured SLOWER here: 272 vs 255 us at B/16 — the
  // second stage halves the workgroups per CU's LDS headroom and the DMA issue competes with the MFMA stream)
  __shared__ __attribute__((aligned(16))) float lds[2 * TS * HD];
  float* Ks = lds;
  float* Vs = lds + TS * HD;
  const int H = a.H, S = a.Sk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int D = H * HD;
  const int q0 = blockIdx.y * 128 + wave * 32;
  const int query = q0 + l31;
  const float* kbase = a.k + (size_t)b * S * a.ldkv + h * HD;
  const float* vbase = a.v + (size_t)b * S * a.ldkv + h * HD;
  // Q fragments (B operand): step (g, r) of the 32 k=2 steps contracts head dims 8g + 4 half + r
  f32x4 qf[8];
  {
    const float* qrow = a.q + ((size_t)b * S + min(query, S - 1)) * a.ldq + h * HD + 4 * half;
#pragma unroll
    for (int g = 0; g < 8; ++g) qf[g] = *reinterpret_cast<const f32x4*>(qrow + 8 * g);
  }
  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  const bool wave_live = q0 < S;                        // a wave past the end only helps staging
  int nkt = (S + TS - 1) / TS;
  if (CAUSAL) nkt = min(nkt, (int)(blockIdx.y * 128 + 127) / TS + 1);
  // The next K / V tile is requested into registers BEFORE the current tile's products (32 VGPRs; the kernel stays at two
  // waves per SIMD) and written to LDS behind them: a tile's memory latency used to sit between two barriers with the
  // matrix pipe idle, once per tile.
  f32x4 nk4[4], nv4[4];
  tile_fetch(nk4, kbase, 0, S, (size_t)a.ldkv);
  tile_fetch(nv4, vbase, 0, S, (size_t)a.ldkv);
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt > 0) __syncthreads();                         // everybody is done reading the previous tile
    tile_commit(Ks, nk4, kt * TS, S);
    tile_commit(Vs, nv4, kt * TS, S);
    __syncthreads();
    if (kt + 1 < nkt) {
      tile_fetch(nk4, kbase, (kt + 1) * TS, S, (size_t)a.ldkv);
      tile_fetch(nv4, vbase, (kt + 1) * TS, S, (size_t)a.ldkv);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!wave_live || (CAUSAL && kt * TS > q0 + 31)) continue;
    // second 32-key half of the tile: skipped when it holds no key this wave needs (wave-uniform)
    const bool sub1 = kt * TS + 32 < S && !(CAUSAL && kt * TS + 32 > q0 + 31);
    f32x16 st[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[sub][r] = 0.f;
      if (sub == 1 && !sub1) continue;
      const int krow = 32 * sub + l31;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        // 16 bytes of key row krow: head dims 8g + 4 half .. +3 (tile_off keeps 4-float slots contiguous)
        const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + krow * HD + (((2 * g + half) ^ (krow & 15)) << 2));
#pragma unroll
        for (int r = 0; r < 4; ++r) st[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[r], qf[g][r], st[sub], 0, 0, 0);
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * TS + 32 * sub + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = st[sub][r] * kScale;
        if (key >= S || (CAUSAL && key > query)) v = -INFINITY;
        st[sub][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);
    const float msafe = (mn == -INFINITY) ? 0.f : mn;
    const float alpha = __expf(m - msafe);
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
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (sub == 1 && !sub1) continue;   // its P is exactly zero (every key masked)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * sub + (r & 3) + 8 * (r >> 2) + 4 * half;   // row of the V tile this lane's k-slot contracts
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[tile_off(key, 32 * dt + l31)], st[sub][r], o[dt], 0, 0, 0);
      }
    }
  }
  if (query < S) {
    const float inv = 1.0f / l;
    float* orow = out + ((size_t)b * S + query) * D + h * HD + 4 * half;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 v = {o[dt][4 * j] * inv, o[dt][4 * j + 1] * inv, o[dt][4 * j + 2] * inv, o[dt][4 * j + 3] * inv};
        *reinterpret_cast<f32x4*>(orow + 32 * dt + 8 * j) = v;
      }
    if (half == 0) lse[(size_t)bh * S + query] = m + __logf(l);
  }
}

// ------------------------------------------------------------------------------------------- backward: dQ
// also writes delta[bh, row] = sum_d dO * O for the dK/dV kernel
template <bool CAUSAL>
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(AttnArgs a, const float* __restrict__ out,
                                                          const float* __restrict__ dout, const float* __restrict__ lse,
                                                          float* __restrict__ dq_out, int lddq, float* __restrict__ delta) {
  const int H = a.H, S = a.Sk, Sq = a.Sq;
  __shared__ __attribute__((aligned(16))) float t0[TS * HD];
  __shared__ __attribute__((aligned(16))) float t1[TS * HD];
  __shared__ __attribute__((aligned(16))) float scratch[4 * 16 * SCR];
  __shared__ float dl_s[TS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qd = lane >> 4, l15 = lane & 15;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int q0 = blockIdx.y * TS;
  const int D = H * HD;
  const float* qbase = a.q + (size_t)b * Sq * a.ldq + h * HD;
  const float* kbase = a.k + (size_t)b * S * a.ldkv + h * HD;
  const float* vbase = a.v + (size_t)b * S * a.ldkv + h * HD;
  const float* obase = out + (size_t)b * Sq * D + h * HD;
  const float* dobase = dout + (size_t)b * Sq * D + h * HD;
  float* scr = scratch + wave * 16 * SCR;

  {  // delta: 4 threads per row, 16 floats each
    const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
    float s = 0.f;
    if (q0 + row < Sq) {
      const f32x4* po = reinterpret_cast<const f32x4*>(obase + (size_t)(q0 + row) * D + part * 16);
      const f32x4* pd = reinterpret_cast<const f32x4*>(dobase + (size_t)(q0 + row) * D + part * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 a = po[i], c = pd[i];
        s += (a[0] * c[0] + a[1] * c[1]) + (a[2] * c[2] + a[3] * c[3]);
      }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (part == 0) {
      dl_s[row] = s;
      if (q0 + row < Sq) delta[(size_t)bh * Sq + q0 + row] = s;
    }
  }
  stage_tiles2(t0, qbase, (size_t)a.ldq, t1, dobase, (size_t)D, q0, Sq);
  __syncthreads();
  f32x4 qf[4], dof[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    qf[g] = frag_k(t0, 16 * wave, g, lane);
    dof[g] = frag_k(t1, 16 * wave, g, lane);
  }
  float lse_r[4], dl[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = q0 + 16 * wave + 4 * qd + r;
    lse_r[r] = (row < Sq) ? lse[(size_t)bh * Sq + row] : 0.f;
    dl[r] = dl_s[16 * wave + 4 * qd + r];
  }
  f32x4 dq[4];
  zero4(dq);
  int nkt = (S + TS - 1) / TS;
  if (CAUSAL) nkt = min(nkt, (int)blockIdx.y + 1);
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    stage_tiles2(t0, kbase, (size_t)a.ldkv, t1, vbase, (size_t)a.ldkv, kt * TS, S);
    __syncthreads();
    f32x4 s[4], dp[4];
    zero4(s);
    zero4(dp);
    mma_rows_x_tileT(s, qf, t0, lane);
    mma_rows_x_tileT(dp, dof, t1, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int key = kt * TS + nt * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qrow = q0 + 16 * wave + 4 * qd + r;
        const bool masked = key >= S || (CAUSAL && key > qrow);
        const float p = masked ? 0.f : __expf(s[nt][r] * kScale - lse_r[r]);
        scr[(4 * qd + r) * SCR + nt * 16 + l15] = p * (dp[nt][r] - dl[r]) * kScale;
      }
    }
    __syncthreads();
    mma_scratch_x_tile(dq, scr, t0, lane);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = q0 + 16 * wave + 4 * qd + r;
    if (row < Sq)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) dq_out[((size_t)b * Sq + row) * lddq + h * HD + nt * 16 + l15] = dq[nt][r];
  }
}

// ------------------------------------------------------------------------------------------- backward: dK, dV
template <bool CAUSAL>
__global__ void __launch_bounds__(256) attn_bwd_dkv_kernel(AttnArgs a, const float* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           float* __restrict__ dk_out, float* __restrict__ dv_out,
                                                           int lddkv) {
  const int H = a.H, S = a.Sk, Sq = a.Sq;
  __shared__ __attribute__((aligned(16))) float t0[TS * HD];
  __shared__ __attribute__((aligned(16))) float t1[TS * HD];
  __shared__ __attribute__((aligned(16))) float scratch_p[4 * 16 * SCR];
  __shared__ __attribute__((aligned(16))) float scratch_s[4 * 16 * SCR];
  __shared__ float lse_s[TS], dl_s[TS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qd = lane >> 4, l15 = lane & 15;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int k0 = blockIdx.y * TS;
  const int D = H * HD;
  const float* qbase = a.q + (size_t)b * Sq * a.ldq + h * HD;
  const float* kbase = a.k + (size_t)b * S * a.ldkv + h * HD;
  const float* vbase = a.v + (size_t)b * S * a.ldkv + h * HD;
  const float* dobase = dout + (size_t)b * Sq * D + h * HD;
  float* scp = scratch_p + wave * 16 * SCR;
  float* scs = scratch_s + wave * 16 * SCR;

  stage_tiles2(t0, kbase, (size_t)a.ldkv, t1, vbase, (size_t)a.ldkv, k0, S);
  __syncthreads();
  f32x4 kf[4], vf[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    kf[g] = frag_k(t0, 16 * wave, g, lane);
    vf[g] = frag_k(t1, 16 * wave, g, lane);
  }
  f32x4 dk[4], dv[4];
  zero4(dk);
  zero4(dv);
  const int nqt = (Sq + TS - 1) / TS;
  for (int qt = CAUSAL ? (int)blockIdx.y : 0; qt < nqt; ++qt) {
    const int q0 = qt * TS;
    __syncthreads();
    stage_tiles2(t0, qbase, (size_t)a.ldq, t1, dobase, (size_t)D, q0, Sq);
    if (threadIdx.x < TS) {
      const int q = q0 + threadIdx.x;
      lse_s[threadIdx.x] = (q < Sq) ? lse[(size_t)bh * Sq + q] : 0.f;
      dl_s[threadIdx.x] = (q < Sq) ? delta[(size_t)bh * Sq + q] : 0.f;
    }
    __syncthreads();
    f32x4 st[4], dpt[4];
    zero4(st);
    zero4(dpt);
    mma_rows_x_tileT(st, kf, t0, lane);    // [key, q] = K Q^T
    mma_rows_x_tileT(dpt, vf, t1, lane);   // [key, q] = V dO^T
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int ql = nt * 16 + l15, q = q0 + ql;
      const float lq = lse_s[ql], dq_ = dl_s[ql];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + 16 * wave + 4 * qd + r;
        const bool masked = key >= S || q >= Sq || (CAUSAL && key > q);
        const float p = masked ? 0.f : __expf(st[nt][r] * kScale - lq);
        scp[(4 * qd + r) * SCR + ql] = p;
        scs[(4 * qd + r) * SCR + ql] = p * (dpt[nt][r] - dq_) * kScale;
      }
    }
    __syncthreads();
    mma_scratch_x_tile(dv, scp, t1, lane);  // dV += P^T dO
    mma_scratch_x_tile(dk, scs, t0, lane);  // dK += dS^T Q
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int key = k0 + 16 * wave + 4 * qd + r;
    if (key < S)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const size_t o = ((size_t)b * S + key) * lddkv + h * HD + nt * 16 + l15;
        dk_out[o] = dk[nt][r];
        dv_out[o] = dv[nt][r];
      }
  }
}

// ------------------------------------------------------------------------------------------- backward, one tile, lean
// S <= 64 (ViT-B/32's 50 tokens), the five-product schedule of attn_bwd_fused_kernel with 48 KiB of LDS instead of
// 80 KiB (three workgroups per CU instead of two: the fused kernel is latency bound at 1.8 waves per SIMD):
//   phase A  wave w owns KEYS 16w..16w+15.  K and V own rows come straight from global as B-operand fragments; the
//            Q and dO tiles are in LDS.  S = Q K^T and dP = dO V^T land as [query 4qd+r][key l15], so P and dS are
//            the B operands of dV^T = dO^T P and dK^T = Q^T dS without leaving registers; dS is also written to
//            an LDS tile for phase B.  dK, dV leave as 16-byte stores.
//   phase B  the K tile is staged over the (now dead) Q tile; wave w owns QUERIES 16w..16w+15 and forms
//            dQ^T = K^T dS^T with dS^T read as 16-byte runs of its own rows of the dS tile.
// IO16: q / k / v, `out`, `dout` and dq / dk / dv are bf16 (see the bf16 I/O note above); lse fp32.
template <bool CAUSAL, bool IO16 = false>
__global__ void __launch_bounds__(256, 3) attn_bwd_lean_kernel(AttnArgs a, const float* __restrict__ out,
                                                               const float* __restrict__ dout, const float* __restrict__ lse,
                                                               float* __restrict__ dq_out, float* __restrict__ dk_out,
                                                               float* __restrict__ dv_out, int ldd) {
  __shared__ __attribute__((aligned(16))) float lds[3 * TS * HD + 2 * TS];
  float* Qs = lds;                 // phase B: the K tile
  float* dOs = lds + TS * HD;
  float* dSs = lds + 2 * TS * HD;
  float* lse_s = lds + 3 * TS * HD;
  float* dl_s = lse_s + TS;
  const int H = a.H, S = a.Sk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qd = lane >> 4, l15 = lane & 15;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int D = H * HD;
  const int own = 16 * wave + l15, ownc = min(own, S - 1);
  const float my_lse = lse[(size_t)bh * S + ownc];      // requested with the rest of the prologue's loads
  f32x4 kf[4], vf[4];
  float my_dl = 0.f;
  f32x4 tk[4];                              // phase B's K tile (fp32 instance): kept in registers through phase A
  u16x8a tk16[2];                           // ... (bf16 instance)
  if constexpr (IO16) {
    const unsigned short* qb = reinterpret_cast<const unsigned short*>(a.q) + (size_t)b * S * a.ldq + h * HD;
    const unsigned short* kb = reinterpret_cast<const unsigned short*>(a.k) + (size_t)b * S * a.ldkv + h * HD;
    const unsigned short* vb = reinterpret_cast<const unsigned short*>(a.v) + (size_t)b * S * a.ldkv + h * HD;
    const unsigned short* dob = reinterpret_cast<const unsigned short*>(dout) + (size_t)b * S * D + h * HD;
    u16x8a tq16[2], tdo16[2];
    tile_fetch16(tq16, qb, S, (size_t)a.ldq);
    tile_fetch16(tdo16, dob, S, (size_t)D);
    tile_fetch16(tk16, kb, S, (size_t)a.ldkv);
    const unsigned short* krow = kb + (size_t)ownc * a.ldkv + 4 * qd;
    const unsigned short* vrow = vb + (size_t)ownc * a.ldkv + 4 * qd;
    const unsigned short* orow = reinterpret_cast<const unsigned short*>(out) + ((size_t)b * S + ownc) * D + h * HD + 4 * qd;
    const unsigned short* drow = dob + (size_t)ownc * D + 4 * qd;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      kf[g] = bf4_to_f32(*reinterpret_cast<const u16x4a*>(krow + 16 * g));
      vf[g] = bf4_to_f32(*reinterpret_cast<const u16x4a*>(vrow + 16 * g));
      const f32x4 o4 = bf4_to_f32(*reinterpret_cast<const u16x4a*>(orow + 16 * g));
      const f32x4 d4 = bf4_to_f32(*reinterpret_cast<const u16x4a*>(drow + 16 * g));
      my_dl += (o4[0] * d4[0] + o4[1] * d4[1]) + (o4[2] * d4[2] + o4[3] * d4[3]);
    }
    __builtin_amdgcn_sched_barrier(0);
    tile_commit16(Qs, tq16, S);
    tile_commit16(dOs, tdo16, S);
#pragma unroll
    for (int c = 0; c < 2; ++c) asm volatile("" ::"v"(tk16[c]));
  } else {
  const float* qbase = a.q + (size_t)b * S * a.ldq + h * HD;
  const float* kbase = a.k + (size_t)b * S * a.ldkv + h * HD;
  const float* vbase = a.v + (size_t)b * S * a.ldkv + h * HD;
  const float* dobase = dout + (size_t)b * S * D + h * HD;
  f32x4 tq[4], tdo[4];                      // Q and dO tiles: requested here, written to LDS behind the fragment loads
  tile_fetch(tq, qbase, 0, S, (size_t)a.ldq);
  tile_fetch(tdo, dobase, 0, S, (size_t)D);
  tile_fetch(tk, kbase, 0, S, (size_t)a.ldkv);
  // own K / V rows as B-operand fragments: row own, head dims 16g + 4qd .. +3
  {
    const float* krow = kbase + (size_t)ownc * a.ldkv + 4 * qd;
    const float* vrow = vbase + (size_t)ownc * a.ldkv + 4 * qd;
    const float* orow = out + ((size_t)b * S + ownc) * D + h * HD + 4 * qd;
    const float* drow = dobase + (size_t)ownc * D + 4 * qd;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      kf[g] = *reinterpret_cast<const f32x4*>(krow + 16 * g);
      vf[g] = *reinterpret_cast<const f32x4*>(vrow + 16 * g);
      const f32x4 o4 = *reinterpret_cast<const f32x4*>(orow + 16 * g);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(drow + 16 * g);
      my_dl += (o4[0] * d4[0] + o4[1] * d4[1]) + (o4[2] * d4[2] + o4[3] * d4[3]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);        // every load of the prologue is in flight before the first wait
  tile_commit(Qs, tq, 0, S);
  tile_commit(dOs, tdo, 0, S);
#pragma unroll
  for (int c = 0; c < 4; ++c) asm volatile("" ::"v"(tk[c]));   // (the optimizer would sink these loads to their use in phase B)
  }
  my_dl += __shfl_xor(my_dl, 16);
  my_dl += __shfl_xor(my_dl, 32);
  if (qd == 0) {
    lse_s[own] = own < S ? my_lse : 0.f;
    dl_s[own] = own < S ? my_dl : 0.f;
  }
  __syncthreads();
  const int kts = (S + 15) / 16;
  // ---- phase A: own keys, every query tile -> dK, dV, and the dS tile
  {
    f32x4 dk[4], dv[4];
    zero4(dk);
    zero4(dv);
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const bool live = qt < kts && (!CAUSAL || qt >= wave);   // wave-uniform
      f32x4 pr = {0.f, 0.f, 0.f, 0.f}, ds = {0.f, 0.f, 0.f, 0.f};
      if (live) {
        f32x4 s2 = {0.f, 0.f, 0.f, 0.f}, dp2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 qf = frag_k(Qs, 16 * qt, g, lane);
          const f32x4 df = frag_k(dOs, 16 * qt, g, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[r], kf[g][r], s2, 0, 0, 0);
            dp2 = __builtin_amdgcn_mfma_f32_16x16x4f32(df[r], vf[g][r], dp2, 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = 16 * qt + 4 * qd + r;
          const bool masked = q >= S || own >= S || (CAUSAL && own > q);
          pr[r] = masked ? 0.f : __expf(s2[r] * kScale - lse_s[q]);
          ds[r] = pr[r] * (dp2[r] - dl_s[q]) * kScale;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = 16 * qt + 4 * qd + r;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            dv[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dOs[tile_off(q, 16 * dt + l15)], pr[r], dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[tile_off(q, 16 * dt + l15)], ds[r], dk[dt], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dSs[tile_off(16 * qt + 4 * qd + r, own)] = ds[r];   // zeros where nothing was computed
    }
    if (own < S) {
      const size_t o = ((size_t)b * S + own) * ldd + h * HD + 4 * qd;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        if constexpr (IO16) {
          *reinterpret_cast<u16x4a*>(reinterpret_cast<unsigned short*>(dk_out) + o + 16 * dt) = f32_to_bf4(dk[dt]);
          *reinterpret_cast<u16x4a*>(reinterpret_cast<unsigned short*>(dv_out) + o + 16 * dt) = f32_to_bf4(dv[dt]);
        } else {
          *reinterpret_cast<f32x4*>(dk_out + o + 16 * dt) = dk[dt];
          *reinterpret_cast<f32x4*>(dv_out + o + 16 * dt) = dv[dt];
        }
      }
    }
  }
  __syncthreads();                                   // Q tile dead, dS tile complete
  if constexpr (IO16) tile_commit16(Qs, tk16, S);    // K over Q
  else tile_commit(Qs, tk, 0, S);
  __syncthreads();
  // ---- phase B: own queries -> dQ^T = K^T dS^T
  {
    const float* Ks = Qs;
    f32x4 dq[4];
    zero4(dq);
    const int n1 = CAUSAL ? min(kts, wave + 1) : kts;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
      if (kt < n1) {
        // dS[own query][keys 16kt + 4qd .. +3]: one 16-byte run of this lane's row of the dS tile
        const f32x4 dst = *reinterpret_cast<const f32x4*>(dSs + own * HD + (((4 * kt + qd) ^ (own & 15)) << 2));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * kt + 4 * qd + r;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
            dq[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[tile_off(key, 16 * dt + l15)], dst[r], dq[dt], 0, 0, 0);
        }
      }
    if (own < S) {
      if constexpr (IO16) {
        unsigned short* o = reinterpret_cast<unsigned short*>(dq_out) + ((size_t)b * S + own) * ldd + h * HD + 4 * qd;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<u16x4a*>(o + 16 * dt) = f32_to_bf4(dq[dt]);
      } else {
        float* o = dq_out + ((size_t)b * S + own) * ldd + h * HD + 4 * qd;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(o + 16 * dt) = dq[dt];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------- backward, streamed
// Long self-attention (Sq == Sk > 80): the two-kernel split of the tiled path (dQ by query blocks; dK, dV by key
// blocks; no atomics) with the register discipline of attn_fwd_stream_kernel — every score / dP block is formed on
// v_mfma_f32_32x32x2_f32 directly in the layout its consumer contracts over, so P and dS never touch LDS.

// dQ for 32 queries per wave: S^T = K Q^T and dP^T = V dO^T (lane = query column), dS^T in registers is the B operand
// of dQ^T = K^T dS^T.  Also writes delta[bh, q] = sum_d O dO for the dK/dV kernel.
template <bool CAUSAL>
__global__ void __launch_bounds__(256) attn_bwd_dq_stream_kernel(AttnArgs a, const float* __restrict__ out,
                                                                 const float* __restrict__ dout, const float* __restrict__ lse,
                                                                 float* __restrict__ dq_out, int lddq, float* __restrict__ delta) {
  __shared__ __attribute__((aligned(16))) float lds[2 * TS * HD];
  float* Ks = lds;
  float* Vs = lds + TS * HD;
  const int H = a.H, S = a.Sk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int D = H * HD;
  const int q0 = blockIdx.y * 128 + wave * 32;
  const int query = q0 + l31, qc = min(query, S - 1);
  const float* kbase = a.k + (size_t)b * S * a.ldkv + h * HD;
  const float* vbase = a.v + (size_t)b * S * a.ldkv + h * HD;
  f32x4 qf[8], dof[8];
  float my_dl = 0.f;
  {
    const float* qrow = a.q + ((size_t)b * S + qc) * a.ldq + h * HD + 4 * half;
    const float* drow = dout + ((size_t)b * S + qc) * D + h * HD + 4 * half;
    const float* orow = out + ((size_t)b * S + qc) * D + h * HD + 4 * half;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      qf[g] = *reinterpret_cast<const f32x4*>(qrow + 8 * g);
      dof[g] = *reinterpret_cast<const f32x4*>(drow + 8 * g);
      const f32x4 o4 = *reinterpret_cast<const f32x4*>(orow + 8 * g);
      my_dl += (o4[0] * dof[g][0] + o4[1] * dof[g][1]) + (o4[2] * dof[g][2] + o4[3] * dof[g][3]);
    }
  }
  my_dl += __shfl_xor(my_dl, 32);
  const float my_lse = lse[(size_t)bh * S + qc];
  if (half == 0 && query < S) delta[(size_t)bh * S + query] = my_dl;
  f32x16 dq[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
  const bool wave_live = q0 < S;
  int nkt = (S + TS - 1) / TS;
  if (CAUSAL) nkt = min(nkt, (int)(blockIdx.y * 128 + 127) / TS + 1);
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt > 0) __syncthreads();
    stage_tiles2(Ks, kbase, (size_t)a.ldkv, Vs, vbase, (size_t)a.ldkv, kt * TS, S);
    __syncthreads();
    if (!wave_live || (CAUSAL && kt * TS > q0 + 31)) continue;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (kt * TS + 32 * sub >= S || (CAUSAL && kt * TS + 32 * sub > q0 + 31)) continue;   // wave-uniform
      f32x16 st, dpt;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[r] = 0.f;
        dpt[r] = 0.f;
      }
      const int krow = 32 * sub + l31;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int off = krow * HD + (((2 * g + half) ^ (krow & 15)) << 2);
        const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + off);
        const f32x4 vf = *reinterpret_cast<const f32x4*>(Vs + off);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[r], qf[g][r], st, 0, 0, 0);
          dpt = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r], dof[g][r], dpt, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * TS + 32 * sub + (r & 3) + 8 * (r >> 2) + 4 * half;
        const bool masked = key >= S || (CAUSAL && key > query);
        const float pr = masked ? 0.f : __expf(st[r] * kScale - my_lse);
        st[r] = pr * (dpt[r] - my_dl) * kScale;   // dS^T
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * sub + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[tile_off(key, 32 * dt + l31)], st[r], dq[dt], 0, 0, 0);
      }
    }
  }
  if (query < S) {
    float* o = dq_out + ((size_t)b * S + query) * lddq + h * HD + 4 * half;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(o + 32 * dt + 8 * j) = f32x4{dq[dt][4 * j], dq[dt][4 * j + 1], dq[dt][4 * j + 2], dq[dt][4 * j + 3]};
  }
}

// dK, dV for 32 keys per wave: S = Q K^T and dP = dO V^T with lane = key column and the queries
// {(r&3) + 8(r>>2) + 4 half} of a 32-query block in registers: P and dS are the B operands of dV^T = dO^T P and
// dK^T = Q^T dS.  Q / dO tiles (64 queries) stream through LDS with their lse and delta.
// WRITE_DS: the dS block (formed here once) also goes to `ds` as dS^T [B*H][Sp keys][Sp queries] (Sp = S rounded up to 32;
// masked entries are written as zeros) for attn_bwd_dq_from_ds_kernel, which then needs ONE product instead of three
// (see the dS-passing note below).
template <bool CAUSAL, bool WRITE_DS = false>
__global__ void __launch_bounds__(256) attn_bwd_dkv_stream_kernel(AttnArgs a, const float* __restrict__ dout,
                                                                  const float* __restrict__ lse, const float* __restrict__ delta,
                                                                  float* __restrict__ dk_out, float* __restrict__ dv_out,
                                                                  int lddkv, float* __restrict__ ds = nullptr, int Sp = 0) {
  __shared__ __attribute__((aligned(16))) float lds[2 * TS * HD + 2 * TS];
  float* Qs = lds;
  float* dOs = lds + TS * HD;
  float* lse_s = lds + 2 * TS * HD;
  float* dl_s = lse_s + TS;
  const int H = a.H, S = a.Sk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int D = H * HD;
  const int k0 = blockIdx.y * 128 + wave * 32;
  const int key = k0 + l31, kc = min(key, S - 1);
  const float* qbase = a.q + (size_t)b * S * a.ldq + h * HD;
  const float* dobase = dout + (size_t)b * S * D + h * HD;
  f32x4 kf[8], vf[8];
  {
    const float* krow = a.k + ((size_t)b * S + kc) * a.ldkv + h * HD + 4 * half;
    const float* vrow = a.v + ((size_t)b * S + kc) * a.ldkv + h * HD + 4 * half;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      kf[g] = *reinterpret_cast<const f32x4*>(krow + 8 * g);
      vf[g] = *reinterpret_cast<const f32x4*>(vrow + 8 * g);
    }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dk[dt][r] = 0.f;
      dv[dt][r] = 0.f;
    }
  const bool wave_live = k0 < S;
  const int nqt = (S + TS - 1) / TS;
  const int qt0 = CAUSAL ? (int)(blockIdx.y * 128) / TS : 0;   // queries before the workgroup's first key see none of its keys
  for (int qt = qt0; qt < nqt; ++qt) {
    if (qt > qt0) __syncthreads();
    stage_tiles2(Qs, qbase, (size_t)a.ldq, dOs, dobase, (size_t)D, qt * TS, S);
    if (threadIdx.x < TS) {
      const int q = qt * TS + threadIdx.x;
      lse_s[threadIdx.x] = q < S ? lse[(size_t)bh * S + q] : 0.f;
      dl_s[threadIdx.x] = q < S ? delta[(size_t)bh * S + q] : 0.f;
    }
    __syncthreads();
    if (!wave_live || (CAUSAL && qt * TS + 63 < k0)) continue;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (qt * TS + 32 * sub >= S || (CAUSAL && qt * TS + 32 * sub + 31 < k0)) continue;   // wave-uniform
      f32x16 s2, dp2;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s2[r] = 0.f;
        dp2[r] = 0.f;
      }
      const int qrow = 32 * sub + l31;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int off = qrow * HD + (((2 * g + half) ^ (qrow & 15)) << 2);
        const f32x4 qf = *reinterpret_cast<const f32x4*>(Qs + off);
        const f32x4 df = *reinterpret_cast<const f32x4*>(dOs + off);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s2 = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[r], kf[g][r], s2, 0, 0, 0);
          dp2 = __builtin_amdgcn_mfma_f32_32x32x2f32(df[r], vf[g][r], dp2, 0, 0, 0);
        }
      }
      f32x16 pr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = 32 * sub + (r & 3) + 8 * (r >> 2) + 4 * half;   // query row inside the tile
        const int q = qt * TS + ql;
        const bool masked = q >= S || key >= S || (CAUSAL && key > q);
        pr[r] = masked ? 0.f : __expf(s2[r] * kScale - lse_s[ql]);
        s2[r] = pr[r] * (dp2[r] - dl_s[ql]) * kScale;   // dS
      }
      if (WRITE_DS) {   // row = this lane's key (< Sp: the wave is live), 4 consecutive queries per 16-byte store
        float* drow = ds + ((size_t)bh * Sp + key) * Sp + qt * TS + 32 * sub + 4 * half;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f32x4*>(drow + 8 * j) = f32x4{s2[4 * j], s2[4 * j + 1], s2[4 * j + 2], s2[4 * j + 3]};
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = 32 * sub + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dOs[tile_off(ql, 32 * dt + l31)], pr[r], dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[tile_off(ql, 32 * dt + l31)], s2[r], dk[dt], 0, 0, 0);
        }
      }
    }
  }
  if (key < S) {
    const size_t o = ((size_t)b * S + key) * lddkv + h * HD + 4 * half;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<f32x4*>(dk_out + o + 32 * dt + 8 * j) = f32x4{dk[dt][4 * j], dk[dt][4 * j + 1], dk[dt][4 * j + 2], dk[dt][4 * j + 3]};
        *reinterpret_cast<f32x4*>(dv_out + o + 32 * dt + 8 * j) = f32x4{dv[dt][4 * j], dv[dt][4 * j + 1], dv[dt][4 * j + 2], dv[dt][4 * j + 3]};
      }
  }
}

// ------------------------------------------------------------------------------------------- backward, streamed, dS passed
// The two-kernel split above forms S and dP twice (7 MFMA products for 5).  With dS passed through memory the dQ kernel
// needs one product: delta first (a row-dot kernel), then attn_bwd_dkv_stream_kernel<.., true> forms S, P, dP, dS ONCE,
// takes dV and dK from them and writes dS^T [B*H][Sp][Sp] (fp32, 308 MB at 128 x 197 tokens x 12 heads — 80 us of HBM time
// against the ~330 us of matrix time the two dropped products took), and attn_bwd_dq_from_ds_kernel computes dQ = dS K.

// delta[bh][q] = sum_d O[q][d] dO[q][d]: 16 lanes per (token, head), 16 pairs per workgroup
__global__ void __launch_bounds__(256) attn_delta_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                                         float* __restrict__ delta, int B, int S, int H) {
  const long pair = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int part = threadIdx.x & 15;
  float s = 0.f;
  const bool live = pair < (long)B * S * H;
  const long token = live ? pair / H : 0;
  const int h = live ? (int)(pair % H) : 0;
  if (live) {
    const size_t off = (size_t)token * H * HD + h * HD + 4 * part;
    const f32x4 o4 = *reinterpret_cast<const f32x4*>(out + off);
    const f32x4 d4 = *reinterpret_cast<const f32x4*>(dout + off);
    s = (o4[0] * d4[0] + o4[1] * d4[1]) + (o4[2] * d4[2] + o4[3] * d4[3]);
  }
  s += __shfl_xor(s, 8);
  s += __shfl_xor(s, 4);
  s += __shfl_xor(s, 2);
  s += __shfl_xor(s, 1);
  if (live && part == 0) {
    const long b = token / S, q = token % S;
    delta[((size_t)b * H + h) * S + q] = s;
  }
}

// dQ^T (64 x 32 queries per wave) = K^T dS^T with dS^T read back from memory: lane (l31, half) loads, for ITS query,
// the keys {(r&3) + 8(r>>2) + 4 half} of a 32-key block — 128-byte runs along the query axis, already the B operand of
// v_mfma_f32_32x32x2_f32 — one block ahead of the products.  The whole K of the head sits in LDS (64-row tiles: 64 KiB
// at 197 tokens, 80 KiB at 257), staged once: no barrier in the loop.
// Measured at 128 x 197 tokens x 12 heads (rocprofv3): 166 us — one product at the ~50 % matrix-pipe efficiency every
// kernel of this family reaches (197 padded to 224 on both axes, one LDS read per MFMA), against 421 us for the kernel
// that re-forms S and dP.  Forms that were slower: dS as [query][key] with 4-byte stores in the dK/dV kernel and a
// per-wave LDS transposition here (149 us, but +80 us of stores there instead of +30); all of a wave's dS^T loads
// requested up front (144-184 VGPRs: 195 us); K streamed through two tiles for four workgroups per CU (205 us).
__global__ void __launch_bounds__(256) attn_bwd_dq_from_ds_kernel(AttnArgs a, const float* __restrict__ ds, int Sp,
                                                                  float* __restrict__ dq_out, int lddq) {
  extern __shared__ __attribute__((aligned(16))) float Kall[];      // [64 * tiles][64], swizzled per 64-row tile
  const int H = a.H, S = a.Sk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int q0 = blockIdx.y * 128 + wave * 32;
  const int query = q0 + l31;
  const float* kbase = a.k + (size_t)b * S * a.ldkv + h * HD;
  const int nkt = (S + TS - 1) / TS;
  for (int kt = 0; kt < nkt; kt += 2) {                            // two tiles' loads in flight
    f32x4 v0[4], v1[4];
    tile_fetch(v0, kbase, kt * TS, S, (size_t)a.ldkv);
    if (kt + 1 < nkt) tile_fetch(v1, kbase, (kt + 1) * TS, S, (size_t)a.ldkv);
    tile_commit(Kall + (size_t)kt * TS * HD, v0, kt * TS, S);
    if (kt + 1 < nkt) tile_commit(Kall + (size_t)(kt + 1) * TS * HD, v1, (kt + 1) * TS, S);
  }
  __syncthreads();
  if (q0 >= S) return;                                             // (after the only barrier)
  // dS^T rows of key block kb (32 keys) for this lane's query: element r at row 32 kb + (r&3) + 8(r>>2) + 4 half
  const float* dcol = ds + (size_t)bh * Sp * Sp + query;           // query < Sp: the wave is live
  const int nkb = (S + 31) / 32;
  f32x16 cur, nxt;
#pragma unroll
  for (int r = 0; r < 16; ++r) cur[r] = dcol[(size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * Sp];
  f32x16 dq[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
  for (int kb = 0; kb < nkb; ++kb) {
    const int kn = min(kb + 1, nkb - 1);                           // (the last iteration re-reads its own block: unused)
#pragma unroll
    for (int r = 0; r < 16; ++r) nxt[r] = dcol[(size_t)(32 * kn + (r & 3) + 8 * (r >> 2) + 4 * half) * Sp];
    const float* Kt = Kall + (size_t)(kb >> 1) * TS * HD;
    const int sub = kb & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 32 * sub + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        dq[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[tile_off(key, 32 * dt + l31)], cur[r], dq[dt], 0, 0, 0);
    }
    cur = nxt;
  }
  if (query < S) {
    float* o = dq_out + ((size_t)b * S + query) * lddq + h * HD + 4 * half;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(o + 32 * dt + 8 * j) = f32x4{dq[dt][4 * j], dq[dt][4 * j + 1], dq[dt][4 * j + 2], dq[dt][4 * j + 3]};
  }
}

// ------------------------------------------------------------------------------------------- backward, one tile
// Self-attention with S <= 64 (ViT-B/32: S = 50): the whole (batch, head) problem is one tile, so S/P/dP/dS are
// formed ONCE and dQ, dK, dV all come out of one workgroup — 5 MFMA products instead of the 7 of the two-kernel
// path, q/k/v/dO read once.  P and dS live in LDS as ordinary swizzled tiles: read by rows (ds_read_b128) they are
// the A operand of dQ = dS K, read by columns (ds_read_b32) they are the A operand of dV = P^T dO and dK = dS^T Q.
// LDS: Q, K, V, dO, P tiles + dS aliased onto V (dead after dP) = 5 x 16 KiB = 80 KiB -> two workgroups per CU.

template <bool CAUSAL>
__global__ void __launch_bounds__(256, 2) attn_bwd_fused_kernel(AttnArgs a, const float* __restrict__ out,
                                                                const float* __restrict__ dout, const float* __restrict__ lse,
                                                                float* __restrict__ dq_out, float* __restrict__ dk_out,
                                                                float* __restrict__ dv_out, int ldd) {
  __shared__ __attribute__((aligned(16))) float lds[5 * TS * HD];
  float* Qs = lds;
  float* Ks = lds + TS * HD;
  float* Vs = lds + 2 * TS * HD;
  float* dOs = lds + 3 * TS * HD;
  float* Ps = lds + 4 * TS * HD;
  float* dSs = Vs;  // V is dead once dP = dO V^T has been formed
  const int H = a.H, S = a.Sk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qd = lane >> 4, l15 = lane & 15;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int D = H * HD;
  const float* qbase = a.q + (size_t)b * S * a.ldq + h * HD;
  const float* kbase = a.k + (size_t)b * S * a.ldkv + h * HD;
  const float* vbase = a.v + (size_t)b * S * a.ldkv + h * HD;
  const float* obase = out + (size_t)b * S * D + h * HD;
  const float* dobase = dout + (size_t)b * S * D + h * HD;

  stage_tiles2(Qs, qbase, (size_t)a.ldq, Ks, kbase, (size_t)a.ldkv, 0, S);
  stage_tiles2(Vs, vbase, (size_t)a.ldkv, dOs, dobase, (size_t)D, 0, S);
  // delta and lse for this lane's 4 query rows (rows 16*wave + 4*qd + r): 16 lanes share a row -> each sums 4 floats
  float dl[4], lse_r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * wave + 4 * qd + r;
    float s = 0.f;
    if (row < S) {
      f32x4 o4 = *reinterpret_cast<const f32x4*>(obase + (size_t)row * D + l15 * 4);
      f32x4 d4 = *reinterpret_cast<const f32x4*>(dobase + (size_t)row * D + l15 * 4);
      s = (o4[0] * d4[0] + o4[1] * d4[1]) + (o4[2] * d4[2] + o4[3] * d4[3]);
    }
    dl[r] = quarter_sum(s);
    lse_r[r] = (row < S) ? lse[(size_t)bh * S + row] : 0.f;
  }
  __syncthreads();
  f32x4 qf[4], dof[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    qf[g] = frag_k(Qs, 16 * wave, g, lane);
    dof[g] = frag_k(dOs, 16 * wave, g, lane);
  }
  f32x4 s[4], dp[4];
  zero4(s);
  zero4(dp);
  mma_rows_x_tileT(s, qf, Ks, lane);
  mma_rows_x_tileT(dp, dof, Vs, lane);
  __syncthreads();  // every wave is done reading V before dS overwrites it
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int key = nt * 16 + l15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qrow = 16 * wave + 4 * qd + r;
      const bool masked = key >= S || qrow >= S || (CAUSAL && key > qrow);
      const float p = masked ? 0.f : __expf(s[nt][r] * kScale - lse_r[r]);
      Ps[tile_off(qrow, key)] = p;
      dSs[tile_off(qrow, key)] = p * (dp[nt][r] - dl[r]) * kScale;
    }
  }
  __syncthreads();
  f32x4 dq[4], dk[4], dv[4];
  zero4(dq);
  zero4(dk);
  zero4(dv);
  mma_tilerows_x_tile(dq, dSs, 16 * wave, Ks, lane);   // dQ[q rows of this wave]   = dS K
  mma_tilecols_x_tile(dv, Ps, 16 * wave, dOs, lane);   // dV[key rows of this wave] = P^T dO
  mma_tilecols_x_tile(dk, dSs, 16 * wave, Qs, lane);   // dK[key rows of this wave] = dS^T Q
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * wave + 4 * qd + r;
    if (row < S)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const size_t o = ((size_t)b * S + row) * ldd + h * HD + nt * 16 + l15;
        dq_out[o] = dq[nt][r];
        dk_out[o] = dk[nt][r];
        dv_out[o] = dv[nt][r];
      }
  }
}

// ------------------------------------------------------------------------------------------- backward, whole rows
// Short self-attention (S <= 16*KT): one workgroup per (batch, head), wave w owns rows 16w..16w+15 both as QUERIES
// (phase 1 -> dQ) and as KEYS (phase 2 -> dK, dV).  Each phase forms the score / dP blocks it needs directly in the
// register layout its products consume (see attn_fwd_rows_kernel: the contraction order is free), so P and dS never
// touch LDS, there is no barrier after the staging one, and dQ / dK / dV leave as 16-byte stores.  The price is
// forming S and dP twice (7 MFMA products instead of 5); the LDS operand traffic drops by a third.
template <int KT, bool CAUSAL>
__global__ void __launch_bounds__(KT * 64) attn_bwd_rows_kernel(AttnArgs a, const float* __restrict__ out,
                                                                const float* __restrict__ dout,
                                                                const float* __restrict__ lse, float* __restrict__ dq_out,
                                                                float* __restrict__ dk_out, float* __restrict__ dv_out,
                                                                int ldd) {
  constexpr int R = KT * 16;
  __shared__ __attribute__((aligned(16))) float lds[4 * R * HD + 2 * R];
  float* Qs = lds;
  float* Ks = lds + R * HD;
  float* Vs = lds + 2 * R * HD;
  float* dOs = lds + 3 * R * HD;
  float* lse_s = lds + 4 * R * HD;
  float* dl_s = lse_s + R;
  const int H = a.H, S = a.Sk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qd = lane >> 4, l15 = lane & 15;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int D = H * HD;
  const float* src[4] = {a.q + (size_t)b * S * a.ldq + h * HD, a.k + (size_t)b * S * a.ldkv + h * HD,
                         a.v + (size_t)b * S * a.ldkv + h * HD, dout + (size_t)b * S * D + h * HD};
  const size_t lds_[4] = {(size_t)a.ldq, (size_t)a.ldkv, (size_t)a.ldkv, (size_t)D};
#pragma unroll
  for (int w2 = 0; w2 < 4; w2 += 2) {            // two tensors (eight loads per thread) in flight at a time
    f32x4 stg[2][4];
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int id = threadIdx.x + c * (KT * 64);
        stg[w][c] = *reinterpret_cast<const f32x4*>(src[w2 + w] + (size_t)min(id >> 4, S - 1) * lds_[w2 + w] + (id & 15) * 4);
      }
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int id = threadIdx.x + c * (KT * 64), row = id >> 4, slot = id & 15;
        if (row >= S) stg[w][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(lds + (w2 + w) * R * HD + row * HD + ((slot ^ (row & 15)) << 2)) = stg[w][c];
      }
  }
  // delta = rowsum(O * dO) and lse of this wave's rows: lane (l15, qd) covers 16 of the 64 head dims of row 16w + l15
  const int own = 16 * wave + l15;
  float my_lse = 0.f, my_dl = 0.f;
  if (own < S) {
    const float* orow = out + ((size_t)b * S + own) * D + h * HD + 4 * qd;
    const float* drow = dout + ((size_t)b * S + own) * D + h * HD + 4 * qd;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 o4 = *reinterpret_cast<const f32x4*>(orow + 16 * g);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(drow + 16 * g);
      my_dl += (o4[0] * d4[0] + o4[1] * d4[1]) + (o4[2] * d4[2] + o4[3] * d4[3]);
    }
    my_lse = lse[(size_t)bh * S + own];
  }
  my_dl += __shfl_xor(my_dl, 16);
  my_dl += __shfl_xor(my_dl, 32);
  if (qd == 0) {
    lse_s[own] = my_lse;
    dl_s[own] = my_dl;
  }
  __syncthreads();
  const int kts = (S + 15) / 16;  // 16-row tiles that hold data

  // ---- phase 1: own queries (lane column l15), every key tile -> dQ
  {
    f32x4 qf[4], dof[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      qf[g] = frag_k(Qs, 16 * wave, g, lane);
      dof[g] = frag_k(dOs, 16 * wave, g, lane);
    }
    f32x4 dq[4];
    zero4(dq);
    const int n1 = CAUSAL ? min(kts, wave + 1) : kts;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
      if (kt < n1) {
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 kf = frag_k(Ks, 16 * kt, g, lane);
          const f32x4 vf = frag_k(Vs, 16 * kt, g, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[r], qf[g][r], st, 0, 0, 0);
            dpt = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r], dof[g][r], dpt, 0, 0, 0);
          }
        }
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * kt + 4 * qd + r;
          const bool masked = key >= S || (CAUSAL && key > own);
          const float pr = masked ? 0.f : __expf(st[r] * kScale - my_lse);
          ds[r] = pr * (dpt[r] - my_dl) * kScale;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * kt + 4 * qd + r;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
            dq[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[tile_off(key, 16 * dt + l15)], ds[r], dq[dt], 0, 0, 0);
        }
      }
    if (own < S) {
      float* o = dq_out + ((size_t)b * S + own) * ldd + h * HD + 4 * qd;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(o + 16 * dt) = dq[dt];
    }
  }

  // ---- phase 2: own keys (lane column l15), every query tile -> dK, dV
  {
    f32x4 kf[4], vf[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      kf[g] = frag_k(Ks, 16 * wave, g, lane);
      vf[g] = frag_k(Vs, 16 * wave, g, lane);
    }
    f32x4 dk[4], dv[4];
    zero4(dk);
    zero4(dv);
#pragma unroll
    for (int qt = 0; qt < KT; ++qt)
      if (qt < kts && (!CAUSAL || qt >= wave)) {
        f32x4 s2 = {0.f, 0.f, 0.f, 0.f}, dp2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 qf = frag_k(Qs, 16 * qt, g, lane);
          const f32x4 df = frag_k(dOs, 16 * qt, g, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[r], kf[g][r], s2, 0, 0, 0);
            dp2 = __builtin_amdgcn_mfma_f32_16x16x4f32(df[r], vf[g][r], dp2, 0, 0, 0);
          }
        }
        f32x4 pr, ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = 16 * qt + 4 * qd + r;
          const bool masked = q >= S || own >= S || (CAUSAL && own > q);
          pr[r] = masked ? 0.f : __expf(s2[r] * kScale - lse_s[q]);
          ds[r] = pr[r] * (dp2[r] - dl_s[q]) * kScale;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = 16 * qt + 4 * qd + r;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            dv[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dOs[tile_off(q, 16 * dt + l15)], pr[r], dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[tile_off(q, 16 * dt + l15)], ds[r], dk[dt], 0, 0, 0);
          }
        }
      }
    if (own < S) {
      const size_t o = ((size_t)b * S + own) * ldd + h * HD + 4 * qd;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *reinterpret_cast<f32x4*>(dk_out + o + 16 * dt) = dk[dt];
        *reinterpret_cast<f32x4*>(dv_out + o + 16 * dt) = dv[dt];
      }
    }
  }
}

}  // namespace

namespace {

template <int KT>
void launch_fwd_rows(const AttnArgs& a, float* out, float* lse, int B, int causal, hipStream_t st) {
  const size_t lds = (size_t)2 * KT * 16 * HD * sizeof(float);
  if (causal) hipLaunchKernelGGL((attn_fwd_rows_kernel<KT, true>), dim3(B * a.H), dim3(KT * 64), lds, st, a, out, lse);
  else hipLaunchKernelGGL((attn_fwd_rows_kernel<KT, false>), dim3(B * a.H), dim3(KT * 64), lds, st, a, out, lse);
}

int launch_fwd(const AttnArgs& a, float* out, float* lse, int B, int causal, hipStream_t st) {
  if (a.Sq == a.Sk && a.Sk <= 80 && !a.q_rows && !getenv("DCLIP_ATTN_TILED")) {   // whole-row kernel (see above)
    switch (cdiv(a.Sk, 16)) {
      case 1: launch_fwd_rows<1>(a, out, lse, B, causal, st); break;
      case 2: launch_fwd_rows<2>(a, out, lse, B, causal, st); break;
      case 3: launch_fwd_rows<3>(a, out, lse, B, causal, st); break;
      case 4: launch_fwd_rows<4>(a, out, lse, B, causal, st); break;
      default: launch_fwd_rows<5>(a, out, lse, B, causal, st); break;
    }
    DCLIP_CHECK_LAUNCH("attention_fwd.rows");
    return DCLIP_OK;
  }
  if (a.Sq == a.Sk && !a.q_rows && !getenv("DCLIP_ATTN_TILED")) {   // long self-attention: streamed, P in registers
    dim3 g2(B * a.H, cdiv(a.Sq, 128));
    if (causal) hipLaunchKernelGGL((attn_fwd_stream_kernel<true>), g2, dim3(256), 0, st, a, out, lse);
    else hipLaunchKernelGGL((attn_fwd_stream_kernel<false>), g2, dim3(256), 0, st, a, out, lse);
    DCLIP_CHECK_LAUNCH("attention_fwd.stream");
    return DCLIP_OK;
  }
  dim3 grid(B * a.H, cdiv(a.Sq, TS)), block(256);
  if (causal) hipLaunchKernelGGL((attn_fwd_kernel<true>), grid, block, 0, st, a, out, lse);
  else hipLaunchKernelGGL((attn_fwd_kernel<false>), grid, block, 0, st, a, out, lse);
  DCLIP_CHECK_LAUNCH("attention_fwd");
  return DCLIP_OK;
}

inline int ds_row_stride(int S) { return cdiv(S, 32) * 32; }
// dS passing: long non-causal self-attention whose whole K fits the dQ kernel's LDS (512 tokens = 128 KiB)
inline bool long_noncausal(int Sq, int Sk, int causal) { return Sq == Sk && Sk > 80 && Sk <= 512 && !causal; }

int launch_bwd(const AttnArgs& a, const float* out, const float* dout, const float* lse, float* dq, int lddq, float* dk,
               float* dv, int lddkv, float* delta, int B, int causal, hipStream_t st, float* ds = nullptr) {
  dim3 gq(B * a.H, cdiv(a.Sq, TS)), gk(B * a.H, cdiv(a.Sk, TS)), block(256);
  if (a.Sq == a.Sk && a.Sk > TS && a.Sk <= 80 && lddq == lddkv && !getenv("DCLIP_ATTN_TILED")) {
    // 65..80 rows (the 77-token text tower when it trains): whole-row kernel, 5 waves.  Measured 214 us vs 362 us for
    // the two-kernel tiled path at B=256, H=8; for S <= 64 the 5-product one-tile kernel below is faster (154 vs 178).
    if (causal)
      hipLaunchKernelGGL((attn_bwd_rows_kernel<5, true>), dim3(B * a.H), dim3(320), 0, st, a, out, dout, lse, dq, dk, dv, lddq);
    else
      hipLaunchKernelGGL((attn_bwd_rows_kernel<5, false>), dim3(B * a.H), dim3(320), 0, st, a, out, dout, lse, dq, dk, dv, lddq);
    DCLIP_CHECK_LAUNCH("attention_bwd.rows");
    return DCLIP_OK;
  }
  if (a.Sq == a.Sk && a.Sq <= TS && lddq == lddkv && !getenv("DCLIP_ATTN_FUSED")) {   // one tile, 48 KiB LDS, P / dS in registers
    if (causal) hipLaunchKernelGGL((attn_bwd_lean_kernel<true>), dim3(B * a.H), block, 0, st, a, out, dout, lse, dq, dk, dv, lddq);
    else hipLaunchKernelGGL((attn_bwd_lean_kernel<false>), dim3(B * a.H), block, 0, st, a, out, dout, lse, dq, dk, dv, lddq);
    DCLIP_CHECK_LAUNCH("attention_bwd.lean");
    return DCLIP_OK;
  }
  if (a.Sq == a.Sk && a.Sq <= TS && lddq == lddkv) {  // one-tile self-attention: fused dQ/dK/dV
    if (causal) hipLaunchKernelGGL((attn_bwd_fused_kernel<true>), dim3(B * a.H), block, 0, st, a, out, dout, lse, dq, dk, dv, lddq);
    else hipLaunchKernelGGL((attn_bwd_fused_kernel<false>), dim3(B * a.H), block, 0, st, a, out, dout, lse, dq, dk, dv, lddq);
    DCLIP_CHECK_LAUNCH("attention_bwd.fused");
    return DCLIP_OK;
  }
  if (a.Sq == a.Sk && !a.q_rows && !causal && !getenv("DCLIP_ATTN_TILED")) {
    // long non-causal self-attention (ViT-B/16, ViT-L/14): streamed kernels, P / dS in registers.  Measured at B/16
    // (128 x 197 tokens, 12 heads) 1160 vs 1486 us, at L/14 1383 vs 1460 us; a causal 130-token case was slower
    // (196 vs 177 us), and no configuration has a causal sequence above 80 tokens, so causal stays on the tiled path.
    dim3 g2(B * a.H, cdiv(a.Sq, 128));
    if (ds && long_noncausal(a.Sq, a.Sk, causal) && !getenv("DCLIP_ATTN_NO_DS")) {   // dS formed once and passed through memory: 5 products instead of 7
      const int Sp = ds_row_stride(a.Sk);
      hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)cdivz((size_t)B * a.Sq * a.H, 16)), block, 0, st, out, dout, delta, B, a.Sq, a.H);
      hipLaunchKernelGGL((attn_bwd_dkv_stream_kernel<false, true>), g2, block, 0, st, a, dout, lse, delta, dk, dv, lddkv, ds, Sp);
      const size_t klds = (size_t)cdiv(a.Sk, TS) * TS * HD * sizeof(float);
      static bool big_lds_set = false;
      if (klds > 64 * 1024 && !big_lds_set) {
        DCLIP_REQUIRE(hipFuncSetAttribute((const void*)attn_bwd_dq_from_ds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          128 * 1024) == hipSuccess, "attention_bwd: cannot raise the dQ kernel's LDS limit");
        big_lds_set = true;
      }
      hipLaunchKernelGGL(attn_bwd_dq_from_ds_kernel, g2, block, klds, st, a, (const float*)ds, Sp, dq, lddq);
      DCLIP_CHECK_LAUNCH("attention_bwd.stream_ds");
      return DCLIP_OK;
    }
    hipLaunchKernelGGL((attn_bwd_dq_stream_kernel<false>), g2, block, 0, st, a, out, dout, lse, dq, lddq, delta);
    hipLaunchKernelGGL((attn_bwd_dkv_stream_kernel<false>), g2, block, 0, st, a, dout, lse, delta, dk, dv, lddkv, (float*)nullptr, 0);
    DCLIP_CHECK_LAUNCH("attention_bwd.stream");
    return DCLIP_OK;
  }
  if (causal) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<true>), gq, block, 0, st, a, out, dout, lse, dq, lddq, delta);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<true>), gk, block, 0, st, a, dout, lse, delta, dk, dv, lddkv);
  } else {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<false>), gq, block, 0, st, a, out, dout, lse, dq, lddq, delta);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<false>), gk, block, 0, st, a, dout, lse, delta, dk, dv, lddkv);
  }
  DCLIP_CHECK_LAUNCH("attention_bwd");
  return DCLIP_OK;
}

}  // namespace

DCLIP_API int dclip_attention_fwd(const float* qkv, float* out, float* lse, int B, int S, int H, int causal,
                                  void* stream) {
  DCLIP_REQUIRE(qkv && out && lse, "attention_fwd: null pointer");
  DCLIP_REQUIRE(B > 0 && S > 0 && H > 0, "attention_fwd: bad shape B=%d S=%d H=%d", B, S, H);
  const int D = H * HD;
  AttnArgs a{qkv, qkv + D, qkv + 2 * D, 3 * D, 3 * D, S, S, H, nullptr};
  return launch_fwd(a, out, lse, B, causal, (hipStream_t)stream);
}

DCLIP_API int dclip_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse,
                                  float* dqkv, float* delta, int B, int S, int H, int causal, void* stream) {
  DCLIP_REQUIRE(qkv && out && dout && lse && dqkv && delta, "attention_bwd: null pointer");
  DCLIP_REQUIRE(B > 0 && S > 0 && H > 0, "attention_bwd: bad shape B=%d S=%d H=%d", B, S, H);
  const int D = H * HD;
  AttnArgs a{qkv, qkv + D, qkv + 2 * D, 3 * D, 3 * D, S, S, H, nullptr};
  return launch_bwd(a, out, dout, lse, dqkv, 3 * D, dqkv + D, dqkv + 2 * D, 3 * D, delta, B, causal, (hipStream_t)stream);
}

// Workspace form: `workspace` holds delta [B*H*S] and, for long non-causal sequences (S > 80: ViT-B/16's 197 tokens,
// ViT-L/14's 257), the dS^T blocks [B*H][Sp][Sp] (Sp = S rounded up to 32) that let the dQ kernel skip re-forming S and dP.
DCLIP_API size_t dclip_attention_bwd_workspace(int B, int S, int H, int causal) {
  size_t n = (size_t)B * H * S;
  n = (n + 63) / 64 * 64;                                   // dS rows start 256-byte aligned
  if (long_noncausal(S, S, causal)) n += (size_t)B * H * ds_row_stride(S) * ds_row_stride(S);
  return n * sizeof(float);
}

DCLIP_API int dclip_attention_bwd_ws(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                                     void* workspace, size_t workspace_bytes, int B, int S, int H, int causal, void* stream) {
  DCLIP_REQUIRE(qkv && out && dout && lse && dqkv && workspace, "attention_bwd_ws: null pointer");
  DCLIP_REQUIRE(B > 0 && S > 0 && H > 0, "attention_bwd_ws: bad shape B=%d S=%d H=%d", B, S, H);
  DCLIP_REQUIRE((uintptr_t)workspace % 16 == 0, "attention_bwd_ws: workspace must be 16-byte aligned");
  if (workspace_bytes < dclip_attention_bwd_workspace(B, S, H, causal)) {
    dclip_set_error("attention_bwd_ws: workspace too small (%zu < %zu)", workspace_bytes,
                    dclip_attention_bwd_workspace(B, S, H, causal));
    return DCLIP_EWORKSPACE;
  }
  const int D = H * HD;
  float* delta = (float*)workspace;
  float* ds = long_noncausal(S, S, causal) ? delta + ((size_t)B * H * S + 63) / 64 * 64 : nullptr;
  AttnArgs a{qkv, qkv + D, qkv + 2 * D, 3 * D, 3 * D, S, S, H, nullptr};
  return launch_bwd(a, out, dout, lse, dqkv, 3 * D, dqkv + D, dqkv + 2 * D, 3 * D, delta, B, causal, (hipStream_t)stream, ds);
}

// bf16 I/O forms for the bf16 training student (S <= 80 forward, S <= 64 backward: the 50-token ViT-B/32): qkv16 [B*S][3D],
// out16 / dout16 [B*S][D], dqkv16 [B*S][3D] are bf16; lse [B*H][S] fp32; arithmetic in fp32 as in the fp32 entry points.
DCLIP_API int dclip_attention_fwd_io16(const void* qkv16, void* out16, float* lse, int B, int S, int H, int causal, void* stream) {
  DCLIP_REQUIRE(qkv16 && out16 && lse, "attention_fwd_io16: null pointer");
  DCLIP_REQUIRE(B > 0 && S > 0 && S <= 80 && H > 0, "attention_fwd_io16: B=%d S=%d (<= 80) H=%d", B, S, H);
  DCLIP_REQUIRE(((uintptr_t)qkv16 | (uintptr_t)out16) % 16 == 0, "attention_fwd_io16: operands must be 16-byte aligned");
  const int D = H * HD;
  const float* q = reinterpret_cast<const float*>(qkv16);          // element pointers are re-typed inside the kernel
  const unsigned short* q16 = reinterpret_cast<const unsigned short*>(qkv16);
  AttnArgs a{q, reinterpret_cast<const float*>(q16 + D), reinterpret_cast<const float*>(q16 + 2 * D), 3 * D, 3 * D, S, S, H, nullptr};
  hipStream_t st = (hipStream_t)stream;
  float* out = reinterpret_cast<float*>(out16);
#define FWD16(KT)                                                                                                             \
  do {                                                                                                                        \
    const size_t lds = (size_t)2 * KT * 16 * HD * sizeof(float);                                                              \
    if (causal) hipLaunchKernelGGL((attn_fwd_rows_kernel<KT, true, true>), dim3(B * H), dim3(KT * 64), lds, st, a, out, lse); \
    else hipLaunchKernelGGL((attn_fwd_rows_kernel<KT, false, true>), dim3(B * H), dim3(KT * 64), lds, st, a, out, lse);       \
  } while (0)
  switch (cdiv(S, 16)) {
    case 1: FWD16(1); break;
    case 2: FWD16(2); break;
    case 3: FWD16(3); break;
    case 4: FWD16(4); break;
    default: FWD16(5); break;
  }
#undef FWD16
  DCLIP_CHECK_LAUNCH("attention_fwd_io16");
  return DCLIP_OK;
}

DCLIP_API int dclip_attention_bwd_io16(const void* qkv16, const void* out16, const void* dout16, const float* lse, void* dqkv16,
                                       int B, int S, int H, int causal, void* stream) {
  DCLIP_REQUIRE(qkv16 && out16 && dout16 && lse && dqkv16, "attention_bwd_io16: null pointer");
  DCLIP_REQUIRE(B > 0 && S > 0 && S <= TS && H > 0, "attention_bwd_io16: B=%d S=%d (<= 64) H=%d", B, S, H);
  DCLIP_REQUIRE(((uintptr_t)qkv16 | (uintptr_t)out16 | (uintptr_t)dout16 | (uintptr_t)dqkv16) % 16 == 0,
                "attention_bwd_io16: operands must be 16-byte aligned");
  const int D = H * HD;
  const unsigned short* q16 = reinterpret_cast<const unsigned short*>(qkv16);
  unsigned short* d16 = reinterpret_cast<unsigned short*>(dqkv16);
  AttnArgs a{reinterpret_cast<const float*>(q16), reinterpret_cast<const float*>(q16 + D), reinterpret_cast<const float*>(q16 + 2 * D),
             3 * D, 3 * D, S, S, H, nullptr};
  hipStream_t st = (hipStream_t)stream;
  const float* o = reinterpret_cast<const float*>(out16);
  const float* dO = reinterpret_cast<const float*>(dout16);
  float* dq = reinterpret_cast<float*>(d16);
  float* dk = reinterpret_cast<float*>(d16 + D);
  float* dv = reinterpret_cast<float*>(d16 + 2 * D);
  if (causal) hipLaunchKernelGGL((attn_bwd_lean_kernel<true, true>), dim3(B * H), dim3(256), 0, st, a, o, dO, lse, dq, dk, dv, 3 * D);
  else hipLaunchKernelGGL((attn_bwd_lean_kernel<false, true>), dim3(B * H), dim3(256), 0, st, a, o, dO, lse, dq, dk, dv, 3 * D);
  DCLIP_CHECK_LAUNCH("attention_bwd_io16");
  return DCLIP_OK;
}

// Last-layer pruning: after the final encoder layer only the CLS row of every image is used (hf:modeling_clip.py:650),
// so the last layer's attention is needed for ONE query row per (image, head) — against all keys.  Same kernels:
// the query "sequence" is row 0 of each image (Sq = 1, batch stride S*3D inside the fused projection).
DCLIP_API int dclip_attention_cls_fwd(const float* qkv, float* out, float* lse, int B, int S, int H, void* stream) {
  DCLIP_REQUIRE(qkv && out && lse, "attention_cls_fwd: null pointer");
  DCLIP_REQUIRE(B > 0 && S > 0 && H > 0, "attention_cls_fwd: bad shape");
  const int D = H * HD;
  AttnArgs a{qkv, qkv + D, qkv + 2 * D, S * 3 * D, 3 * D, 1, S, H, nullptr};
  return launch_fwd(a, out, lse, B, 0, (hipStream_t)stream);
}

// dqkv must be zero-filled by the caller where this call does not write (the q part of every non-CLS row).
DCLIP_API int dclip_attention_cls_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                                      float* delta, int B, int S, int H, void* stream) {
  DCLIP_REQUIRE(qkv && out && dout && lse && dqkv && delta, "attention_cls_bwd: null pointer");
  DCLIP_REQUIRE(B > 0 && S > 0 && H > 0, "attention_cls_bwd: bad shape");
  const int D = H * HD;
  AttnArgs a{qkv, qkv + D, qkv + 2 * D, S * 3 * D, 3 * D, 1, S, H, nullptr};
  return launch_bwd(a, out, dout, lse, dqkv, S * 3 * D, dqkv + D, dqkv + 2 * D, 3 * D, delta, B, 0, (hipStream_t)stream);
}

// Text tower, last layer, frozen: only the first-EOS row of each caption is pooled (hf:modeling_clip.py:574-581), so
// the attention output is needed for that one row, against keys 0..eos[b] (causal).  out [B, H*64].
DCLIP_API int dclip_attention_row_fwd(const float* qkv, const int32_t* rows, float* out, float* lse, int B, int S, int H,
                                      void* stream) {
  DCLIP_REQUIRE(qkv && rows && out && lse, "attention_row_fwd: null pointer");
  DCLIP_REQUIRE(B > 0 && S > 0 && H > 0, "attention_row_fwd: bad shape");
  const int D = H * HD;
  AttnArgs a{qkv, qkv + D, qkv + 2 * D, 3 * D, 3 * D, 1, S, H, rows};
  return launch_fwd(a, out, lse, B, 0, (hipStream_t)stream);
}

DCLIP_API int dclip_cross_attention_fwd(const float* q, const float* kv, float* out, float* lse, int B, int Lq, int Lk,
                                        int H, void* stream) {
  DCLIP_REQUIRE(q && kv && out && lse, "cross_attention_fwd: null pointer");
  DCLIP_REQUIRE(B > 0 && Lq > 0 && Lk > 0 && H > 0, "cross_attention_fwd: bad shape B=%d Lq=%d Lk=%d H=%d", B, Lq, Lk, H);
  const int E = H * HD;
  AttnArgs a{q, kv, kv + E, E, 2 * E, Lq, Lk, H, nullptr};
  return launch_fwd(a, out, lse, B, 0, (hipStream_t)stream);
}

DCLIP_API int dclip_cross_attention_bwd(const float* q, const float* kv, const float* out, const float* dout,
                                        const float* lse, float* dq, float* dkv, float* delta, int B, int Lq, int Lk,
                                        int H, void* stream) {
  DCLIP_REQUIRE(q && kv && out && dout && lse && dq && dkv && delta, "cross_attention_bwd: null pointer");
  DCLIP_REQUIRE(B > 0 && Lq > 0 && Lk > 0 && H > 0, "cross_attention_bwd: bad shape");
  const int E = H * HD;
  AttnArgs a{q, kv, kv + E, E, 2 * E, Lq, Lk, H, nullptr};
  return launch_bwd(a, out, dout, lse, dq, E, dkv, dkv + E, 2 * E, delta, B, 0, (hipStream_t)stream);
}
