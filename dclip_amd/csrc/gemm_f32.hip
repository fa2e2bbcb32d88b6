// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD).
//
// C[M,N] = epilogue( alpha * sum_k A(m,k) B(k,n) ),  either operand K-major or M/N-major.
//
// Tiling (one workgroup = 4 waves = 256 lanes):
//   block tile  BM x BN x 32,  waves in a WM x WN grid, each wave owns (BM/WM) x (BN/WN) as
//   MT x NT MFMA tiles of 32x32 (16 accumulator registers each).
// LDS images (double buffered, one barrier per K-tile):
//   K-major operand  [rows][32 floats] = 128-B rows, 16-B slots XOR-swizzled with (row>>1)&7 so that a
//     ds_read_b128 of one slot down 16 rows touches 16 distinct slots of the 256-B bank row.  A lane
//     (row = lane&31, half = lane>>5) reads slot 2g+half of k-group g and gets k = 8g + 4*half + {0..3}.
//   M/N-major operand [32][rows]: a lane reads element (k, row0 + (lane&31)) with ds_read_b32 —
//     32 consecutive floats per half-wave, conflict free; the same k = 8g + 4*half + r is chosen.
//   The MFMA only needs A and B to agree on which k each lane-half supplies, so the k permutation is free.
// Global -> LDS goes through registers (global_load_dwordx4, zero-filled out of range) one K-tile ahead.
// Workgroup ids are remapped so that the 8 XCDs each walk a contiguous band of row-tiles (A panel reuse in
// that XCD's L2; B, the weight, is small and shared through the Infinity Cache).
#include "common.h"
#include <stdlib.h>

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

constexpr int BK = 32;

struct GemmParams {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* residual;
  float* aux;
  int M, N, K;
  int lda, ldb, ldc;
  int epilogue;
  float alpha;
  int tiles_m, tiles_n;
  int k_per_split;  // multiple of BK
  float* slab;      // split-K partials [split][M][N] or nullptr
  float* rs_slab;   // A_ROWSUM under split-K: partials [split][M]
  int group_m;      // tile rows per rasterisation group (an XCD's ~96 co-resident tiles cover a compact patch)
  // contrastive-loss modes (mode 0 = plain GEMM)
  int mode;              // 1: row-LSE partials over this tile's columns, 2: dZ tile, 3: per-row count of z > lse_row[row]
  const float* lse_row;  // mode 2
  const float* lse_col;  // mode 2
  float* part_m;         // mode 1: [tiles_n*WN][M] running max
  float* part_s;         // mode 1: [tiles_n*WN][M] sum exp(z - max)
  int offset;            // column of row i's positive = i + offset
  const int32_t* gt;     // mode 3: per-row ground-truth column, excluded from the count (nullptr = row index)
  unsigned magic_per_group;   // floor(2^32 / (group_m * tiles_n)), filled in by launch_cfg
#ifdef DCLIP_GEMM_STAMPS
  int dbg;               // diagnostic ablations (WRONG results): 1 no epilogue, 2 no DMA in the K loop, 4 no K-loop barrier,
                         // 8 DMA pieces fetched out of range (issue + zero fill, no memory traffic)
#endif
};

enum { MODE_GEMM = 0, MODE_LSE = 1, MODE_DZ = 2, MODE_RANK = 3 };

// Diagnostic build only (`make stamps` -> tools/ab/libdclip_hip_stamps.so, never the product library): per workgroup,
// shader-clock stamps at kernel entry / first barrier / end of the K loop / exit, the 100 MHz real-time clock at entry
// and exit (comparable across CUs), where it ran, and wave 0's cycles in the end-of-tile vmcnt wait and barrier.
// Stamp values go to a buffer of their own.  tools/gemm_stamps.py, tools/gemm_ablate.py.
#ifdef DCLIP_GEMM_STAMPS
__device__ unsigned long long* g_stamps = nullptr;
#define GEMM_STAMP_V(slot, value)                                                              \
  do {                                                                                         \
    if (threadIdx.x == 0 && g_stamps)                                                          \
      g_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (slot)] = (value);         \
  } while (0)
#define GEMM_STAMP(slot) GEMM_STAMP_V(slot, __builtin_amdgcn_s_memtime())
#define GEMM_STAMP_RT(slot) GEMM_STAMP_V(slot, __builtin_amdgcn_s_memrealtime())
#define GEMM_DBG(bit) (p.dbg & (bit))
#else
#define GEMM_STAMP_V(slot, value) do {} while (0)
#define GEMM_STAMP(slot) do {} while (0)
#define GEMM_STAMP_RT(slot) do {} while (0)
#define GEMM_DBG(bit) false
#endif

__device__ __forceinline__ void apply_epilogue_store(const GemmParams& p, int row, int col, float v) {
  v *= p.alpha;
  size_t off = (size_t)row * p.ldc + col;
  if (p.epilogue & DCLIP_EPI_BIAS) v += p.bias[col];
  if (p.epilogue & DCLIP_EPI_GELU) {
    if (p.aux) p.aux[off] = v;
    v = quick_gelu_f(v);
  }
  if (p.epilogue & DCLIP_EPI_DGELU) v *= quick_gelu_grad_f(p.aux[off]);
  if (p.epilogue & DCLIP_EPI_RESIDUAL) v += p.residual[off];
  if (p.epilogue & DCLIP_EPI_ACCUM) v += p.C[off];
  p.C[off] = v;
}

// XCD-aware, bijective remap of the linear workgroup id (blocks b and b+8 share an XCD).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + i;
}

// 16 bytes per lane, global -> LDS without passing through registers (lane i lands at lds_dst + 16 i).  A __device__
// function on purpose: with the builtin in a lambda of the kernel template hipcc 7.2 drops the template's host stub.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds_dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}

template <int BM, int BN, int WM, int WN, bool A_KMAJOR, bool B_KMAJOR, bool DMA = false>
__global__ void __launch_bounds__(WM * WN * 64, 2) gemm_f32_kernel(GemmParams p) {
  constexpr int NTHR = WM * WN * 64;  // 4 waves, or 8 for the 128x128 tile (two 64x32 wave tiles per SIMD)
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MT = TM / 32, NT = TN / 32;
  constexpr int A_CHUNKS = BM * BK / 4 / NTHR;  // 16-byte chunks per thread per K-tile
  constexpr int B_CHUNKS = BN * BK / 4 / NTHR;
  static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
  static_assert(A_CHUNKS >= 1 && B_CHUNKS >= 1, "tile too small");

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* As = lds;                      // [2][BM*BK]
  float* Bs = lds + 2 * BM * BK;        // [2][BN*BK]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, half = lane >> 5;

  GEMM_STAMP(0);
  GEMM_STAMP_RT(4);
  // (A wave that is NOT in its K loop shares its SIMD with waves issuing an MFMA every 64 cycles and gets an instruction
  // through only every 50-200 cycles — tools/gemm_stamps.py: ~14 k cycles of prologue and ~13 k of epilogue around a
  // 145 k-cycle K loop at K = 768.  Raising its priority with s_setprio 3 outside the K loop changed nothing (A/B within
  // +-0.5 % on every shape), so what is left is to keep prologue and epilogue SHORT in instructions.)
  const int nwg = p.tiles_m * p.tiles_n;
  // Each XCD walks a contiguous range of `swz`; inside it tiles are visited in groups of GROUP_M tile-rows, column
  // by column, so the ~64 workgroups resident on one XCD cover an ~8x8 patch of tiles: every A / B panel slice
  // fetched into that XCD's L2 is shared by 8 tiles (fabric reads ~ |A|*tiles_n/8 + |B|*tiles_m/8).
  const int GROUP_M = p.group_m;
  const int swz = xcd_remap(blockIdx.x, nwg);
  const int per_group = GROUP_M * p.tiles_n;
  // swz / per_group by the host's floor(2^32 / per_group): a scalar multiply-high and one correction instead of an
  // integer-division sequence (every instruction of the prologue costs tens of cycles beside MFMA-issuing waves)
  int grp = (int)__umulhi((unsigned)swz, p.magic_per_group);
  int rem = swz - grp * per_group;
  if (rem >= per_group) {
    rem -= per_group;
    ++grp;
  }
  const int first_m = grp * GROUP_M;
  const int gsize = min(GROUP_M, p.tiles_m - first_m);
  int tile_m, tile_n;
  if (gsize == 4) {             // every full group of the default rasterisation
    tile_m = first_m + (rem & 3);
    tile_n = rem >> 2;
  } else if (gsize == 8) {
    tile_m = first_m + (rem & 7);
    tile_n = rem >> 3;
  } else {
    tile_m = first_m + rem % gsize;
    tile_n = rem / gsize;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg = blockIdx.y * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nk = (kend - kbeg + BK - 1) / BK;

  f32x16 acc[MT][NT];   // zeroed BEHIND the first tile's loads (below): those v_movs then cost nothing

  // DCLIP_EPI_A_ROWSUM: sum_k A[m][k] is accumulated from the staged chunks of a [K][M]-major A while they sit in
  // registers (every chunk of a thread covers the same 4 columns m: 256 % (BM/4) == 0); the workgroups of tile
  // column 0 reduce across threads through LDS after the K loop and store.
  static_assert(NTHR % (BM / 4) == 0, "a thread's A chunks must share their m columns");
  const bool do_rs = !A_KMAJOR && (p.epilogue & DCLIP_EPI_A_ROWSUM) && tile_n == 0;   // workgroup-uniform
  f32x4 rs4 = {0.f, 0.f, 0.f, 0.f};

  f32x4 ra[A_CHUNKS], rb[B_CHUNKS];

  // ---- staging through buffer loads -------------------------------------------------------------------------
  // Each operand gets one buffer descriptor (SRD) whose base is this workgroup's tile origin; every staged
  // 16-byte chunk has a per-lane byte offset computed ONCE, and walking K only changes the scalar offset — no
  // vector ALU work per K-tile.  Rows past M/N are clamped onto the last valid row (they only feed output rows /
  // columns that are never stored).  For an [K][M]-major operand rows past kend fall outside the descriptor and
  // read as zero; for a K-major operand a ragged K tail (K % 32 != 0) is zeroed with selects in the last tile only.
  const float* a_org = A_KMAJOR ? p.A + (size_t)m0 * p.lda + kbeg : p.A + (size_t)kbeg * p.lda + m0;
  const float* b_org = B_KMAJOR ? p.B + (size_t)n0 * p.ldb + kbeg : p.B + (size_t)kbeg * p.ldb + n0;
  const int a_rows = min(BM, p.M - m0), b_rows = min(BN, p.N - n0);
  const int kspan = kend - kbeg;  // > 0
  const size_t a_bytes = A_KMAJOR ? ((size_t)(a_rows - 1) * p.lda + (p.K - kbeg)) * 4
                                  : ((size_t)(kspan - 1) * p.lda + (p.M - m0)) * 4;
  const size_t b_bytes = B_KMAJOR ? ((size_t)(b_rows - 1) * p.ldb + (p.K - kbeg)) * 4
                                  : ((size_t)(kspan - 1) * p.ldb + (p.N - n0)) * 4;
  const __amdgpu_buffer_rsrc_t a_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_org), 0, (int)min(a_bytes, (size_t)0x7fffffff), 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b_org), 0, (int)min(b_bytes, (size_t)0x7fffffff), 0x00020000);
  const int a_kstep = A_KMAJOR ? BK * 4 : BK * p.lda * 4;  // scalar byte advance per K-tile
  const int b_kstep = B_KMAJOR ? BK * 4 : BK * p.ldb * 4;
  int a_voff[A_CHUNKS], b_voff[B_CHUNKS];
#pragma unroll
  for (int c = 0; c < A_CHUNKS; ++c) {
    const int id = tid + c * NTHR;
    if (A_KMAJOR) a_voff[c] = (min(id >> 3, a_rows - 1) * p.lda + (id & 7) * 4) * 4;
    else a_voff[c] = ((id / (BM / 4)) * p.lda + min((id % (BM / 4)) * 4, p.M - m0 - 4)) * 4;
  }
#pragma unroll
  for (int c = 0; c < B_CHUNKS; ++c) {
    const int id = tid + c * NTHR;
    if (B_KMAJOR) b_voff[c] = (min(id >> 3, b_rows - 1) * p.ldb + (id & 7) * 4) * 4;
    else b_voff[c] = ((id / (BN / 4)) * p.ldb + min((id % (BN / 4)) * 4, p.N - n0 - 4)) * 4;
  }

  // LDS-DMA staging (DMA variants, steady state only): chunk `id` must LAND at linear LDS position id, so for a
  // swizzled K-major image the lane fetches the logical slot that lives there: (id & 7) ^ ((row >> 1) & 7).
  int a_dvoff[DMA ? A_CHUNKS : 1], b_dvoff[DMA ? B_CHUNKS : 1];
  if (DMA) {
#pragma unroll
    for (int c = 0; c < A_CHUNKS; ++c) {
      const int id = tid + c * NTHR, row = id >> 3;
      a_dvoff[c] = A_KMAJOR ? (min(row, a_rows - 1) * p.lda + (((id & 7) ^ ((row >> 1) & 7)) * 4)) * 4 : a_voff[c];
    }
#pragma unroll
    for (int c = 0; c < B_CHUNKS; ++c) {
      const int id = tid + c * NTHR, row = id >> 3;
      b_dvoff[c] = B_KMAJOR ? (min(row, b_rows - 1) * p.ldb + (((id & 7) ^ ((row >> 1) & 7)) * 4)) * 4 : b_voff[c];
    }
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto dma_chunk = [&](int buf, int kt, int c) {   // chunk c of tile kt -> stage `buf` (A chunks first, then B)
    if (c < A_CHUNKS) dma16(a_rsrc, As + buf * BM * BK + (wave_u + (NTHR / 64) * c) * 256, a_dvoff[DMA ? c : 0], kt * a_kstep);
    else
      dma16(b_rsrc, Bs + buf * BN * BK + (wave_u + (NTHR / 64) * (c - A_CHUNKS)) * 256, b_dvoff[DMA ? c - A_CHUNKS : 0],
            kt * b_kstep);
  };

  auto load_tile = [&](int kt) {
#pragma unroll
    for (int c = 0; c < A_CHUNKS; ++c)
      ra[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff[c], kt * a_kstep, 0));
#pragma unroll
    for (int c = 0; c < B_CHUNKS; ++c)
      rb[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_voff[c], kt * b_kstep, 0));
  };

  // registers -> LDS.  `ragged`: this is the last, partial K-tile of a K-major operand: zero k >= kend.
  auto store_tile = [&](int buf, int kt, bool ragged) {
    float* a = As + buf * BM * BK;
    float* b = Bs + buf * BN * BK;
    const int krem = kspan - kt * BK;  // valid k in this tile (>= BK unless ragged)
#pragma unroll
    for (int c = 0; c < A_CHUNKS; ++c) {
      const int id = tid + c * NTHR;
      f32x4 v = ra[c];
      if (A_KMAJOR) {
        const int row = id >> 3, slot = id & 7;
        if (ragged) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (slot * 4 + e < krem) ? v[e] : 0.f;
        }
        *reinterpret_cast<f32x4*>(a + row * BK + ((slot ^ ((row >> 1) & 7)) << 2)) = v;
      } else {
        *reinterpret_cast<f32x4*>(a + id * 4) = v;  // [kk][BM] is exactly chunk order
        rs4 += v;
      }
    }
#pragma unroll
    for (int c = 0; c < B_CHUNKS; ++c) {
      const int id = tid + c * NTHR;
      f32x4 v = rb[c];
      if (B_KMAJOR) {
        const int row = id >> 3, slot = id & 7;
        if (ragged) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (slot * 4 + e < krem) ? v[e] : 0.f;
        }
        *reinterpret_cast<f32x4*>(b + row * BK + ((slot ^ ((row >> 1) & 7)) << 2)) = v;
      } else {
        *reinterpret_cast<f32x4*>(b + id * 4) = v;
      }
    }
  };

  auto read_frags = [&](const float* a, const float* b, int g, f32x4 (&fa)[MT], f32x4 (&fb)[NT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = wm * TM + i * 32 + l31;
      if (A_KMAJOR) {
        fa[i] = *reinterpret_cast<const f32x4*>(a + row * BK + (((2 * g + half) ^ ((row >> 1) & 7)) << 2));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) fa[i][r] = a[(8 * g + 4 * half + r) * BM + row];
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int row = wn * TN + j * 32 + l31;
      if (B_KMAJOR) {
        fb[j] = *reinterpret_cast<const f32x4*>(b + row * BK + (((2 * g + half) ^ ((row >> 1) & 7)) << 2));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) fb[j][r] = b[(8 * g + 4 * half + r) * BN + row];
      }
    }
  };

  // fragments of k-group g+1 are read from LDS before the 4*MT*NT MFMAs of group g are issued
  auto compute_tile = [&](int buf) {
    const float* a = As + buf * BM * BK;
    const float* b = Bs + buf * BN * BK;
    f32x4 fa[2][MT], fb[2][NT];
    read_frags(a, b, 0, fa[0], fb[0]);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g + 1 < 4) read_frags(a, b, g + 1, fa[(g + 1) & 1], fb[(g + 1) & 1]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][i][r], fb[g & 1][j][r], acc[i][j], 0, 0, 0);
    }
  };

  // unmasked registers -> LDS for ONE staged chunk (A chunks first, then B chunks)
  auto store_chunk = [&](int buf, int c) {
    if (c < A_CHUNKS) {
      const int id = tid + c * NTHR;
      float* a = As + buf * BM * BK;
      if (A_KMAJOR) {
        const int row = id >> 3, slot = id & 7;
        *reinterpret_cast<f32x4*>(a + row * BK + ((slot ^ ((row >> 1) & 7)) << 2)) = ra[c];
      } else {
        *reinterpret_cast<f32x4*>(a + id * 4) = ra[c];
      }
    } else {
      const int cb = c - A_CHUNKS;
      const int id = tid + cb * NTHR;
      float* b = Bs + buf * BN * BK;
      if (B_KMAJOR) {
        const int row = id >> 3, slot = id & 7;
        *reinterpret_cast<f32x4*>(b + row * BK + ((slot ^ ((row >> 1) & 7)) << 2)) = rb[cb];
      } else {
        *reinterpret_cast<f32x4*>(b + id * 4) = rb[cb];
      }
    }
  };

  // Steady state (the next K-tile exists and is full): its global loads are issued behind the first fragment
  // reads, and its LDS writes are spread one chunk per MFMA step over the second half of the running tile, so that
  // after the tile's last MFMA only the barrier remains.
  auto compute_and_stage = [&](int buf, int kt) {
    const float* a = As + buf * BM * BK;
    const float* b = Bs + buf * BN * BK;
    constexpr int NCH = A_CHUNKS + B_CHUNKS;
    constexpr int FIRST = 14 - NCH;  // MFMA steps FIRST .. FIRST+NCH-1 each carry one chunk store
    f32x4 fa[2][MT], fb[2][NT];
    read_frags(a, b, 0, fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);  // first fragments first: the MFMAs wait on them, not on the prefetch issue
    load_tile(kt + 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch at the top: hipcc otherwise sinks the loads to their uses
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g + 1 < 4) read_frags(a, b, g + 1, fa[(g + 1) & 1], fb[(g + 1) & 1]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][i][r], fb[g & 1][j][r], acc[i][j], 0, 0, 0);
        const int step = g * 4 + r;
        if (step >= FIRST && step - FIRST < NCH) {
          store_chunk(buf ^ 1, step - FIRST);
          __builtin_amdgcn_sched_barrier(0);  // one LDS write per MFMA step, in this order
        }
        if (!A_KMAJOR && step == FIRST + NCH) {
          // every staged chunk has landed and is in LDS: fold the A chunks into the row sums BEHIND this step's
          // MFMAs (placed ahead of an MFMA, even four independent adds delay its issue and cost ~6 % of the tile)
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < A_CHUNKS; ++c) rs4 += ra[c];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };

  // DMA variant of the steady state: no staging registers, no ds_write (the VGPR -> LDS transfer path moves only
  // ~79 B/clk/CU and is shared by the three co-resident workgroups) — one DMA piece behind each of MFMA steps
  // 1..NCH (so every piece has >= 16 - NCH steps to land), then vmcnt(0) and the barrier.  Issuing on every other
  // step instead measured +0.3 % rather than +1.1 %.
#ifdef DCLIP_GEMM_STAMPS
  unsigned long long st_vm = 0, st_bar = 0;   // cycles wave 0 spent in the end-of-tile vmcnt(0) wait / in the barrier
#endif
  auto compute_and_dma = [&](int buf, int kt) {
    const float* a = As + buf * BM * BK;
    const float* b = Bs + buf * BN * BK;
    constexpr int NCH = A_CHUNKS + B_CHUNKS;
    f32x4 fa[2][MT], fb[2][NT];
    read_frags(a, b, 0, fa[0], fb[0]);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g + 1 < 4) read_frags(a, b, g + 1, fa[(g + 1) & 1], fb[(g + 1) & 1]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][i][r], fb[g & 1][j][r], acc[i][j], 0, 0, 0);
        const int step = g * 4 + r;
        if (step >= 1 && step - 1 < NCH) {
          __builtin_amdgcn_sched_barrier(0);
          if (GEMM_DBG(8)) dma_chunk(buf ^ 1, 0x3fffff, step - 1);
          else if (!GEMM_DBG(2)) dma_chunk(buf ^ 1, kt + 1, step - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#ifdef DCLIP_GEMM_STAMPS
    const unsigned long long w0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    st_vm += __builtin_amdgcn_s_memtime() - w0;
#else
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces of the next tile have landed
#endif
  };

  const bool k_ragged = (A_KMAJOR || B_KMAJOR) && (kspan % BK) != 0;  // only the last tile can be partial
  GEMM_STAMP(8);                       // setup done (descriptors, offsets)
  // First K-tile.  The LDS-DMA kernels fetch it by DMA as well when it is a full tile (no staging registers, no
  // ds_write); either way the accumulators are zeroed while the loads are in flight.
  const bool dma0 = DMA && !do_rs && !(k_ragged && nk == 1);
  if (dma0) {
#pragma unroll
    for (int c = 0; c < A_CHUNKS + B_CHUNKS; ++c) dma_chunk(0, 0, c);
  } else {
    load_tile(0);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  __builtin_amdgcn_sched_barrier(0);
  GEMM_STAMP(9);
  if (dma0) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  } else {
    if (k_ragged && nk == 1) store_tile(0, 0, true);
    else store_tile(0, 0, false);
  }
  GEMM_STAMP(10);
  __syncthreads();
  GEMM_STAMP(1);
  int kt = 0;
  const int n_steady = nk - 1 - (k_ragged ? 1 : 0);  // tiles whose successor is a full tile
  for (; kt < n_steady; ++kt) {
    if (DMA && !do_rs) compute_and_dma(kt & 1, kt);   // the row sums need the chunks in registers: tile column 0 keeps them
    else compute_and_stage(kt & 1, kt);
#ifdef DCLIP_GEMM_STAMPS
    const unsigned long long b0 = __builtin_amdgcn_s_memtime();
    if (!GEMM_DBG(4)) __syncthreads();
    st_bar += __builtin_amdgcn_s_memtime() - b0;
#else
    __syncthreads();
#endif
  }
  for (; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    compute_tile(buf);
    if (kt + 1 < nk) {
      if (k_ragged && kt + 2 == nk) store_tile(buf ^ 1, kt + 1, true);
      else store_tile(buf ^ 1, kt + 1, false);
    }
    __syncthreads();
  }
  GEMM_STAMP(2);
#ifdef DCLIP_GEMM_STAMPS
  {
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));    // HW_REG_XCC_ID[3:0]
    GEMM_STAMP_V(6, ((unsigned long long)xcc << 32) | hw);
    // slot 7: tile index (20 bits) | vmcnt-wait cycles (22 bits) | barrier-wait cycles (22 bits), both saturating
    const unsigned long long vmc = st_vm > 0x3FFFFF ? 0x3FFFFF : st_vm, brc = st_bar > 0x3FFFFF ? 0x3FFFFF : st_bar;
    GEMM_STAMP_V(7, (unsigned long long)(swz & 0xFFFFF) | (vmc << 20) | (brc << 42));
  }
  if (GEMM_DBG(1)) {          // no epilogue at all: keep the accumulators alive with a store that never executes
    if (p.alpha == 12345.678f) p.C[threadIdx.x] = acc[0][0][0] + acc[MT - 1][NT - 1][15];
    GEMM_STAMP(3);
    GEMM_STAMP_RT(5);
    return;
  }
#endif

  if (do_rs) {   // all LDS reads of the K loop are behind the last barrier: reuse the tile space
    *reinterpret_cast<f32x4*>(lds + tid * 4) = rs4;
    __syncthreads();
    if (tid < BM && m0 + tid < p.M) {
      constexpr int MC = BM / 4;  // m-chunks per K row; thread t holds chunk t % MC
      float t = 0.f;
#pragma unroll
      for (int kr = 0; kr < NTHR / MC; ++kr) t += lds[(kr * MC + tid / 4) * 4 + (tid & 3)];
      if (p.slab) p.rs_slab[(size_t)blockIdx.y * p.M + m0 + tid] = t;
      else p.aux[m0 + tid] = t;
    }
    __syncthreads();
  }

  if (p.mode == MODE_LSE) {
    // z = alpha * acc (alpha = 1/temperature).  Per row: max and sum-exp over this wave's TN columns,
    // reduced over the 32 lanes that share a row with xor-shuffles; partial slot = tile_n*WN + wn.
    const int slot = tile_n * WN + wn;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v[NT], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int col = n0 + wn * TN + j * 32 + l31;
          v[j] = (col < p.N) ? acc[i][j][r] * p.alpha : -INFINITY;
          mx = fmaxf(mx, v[j]);
        }
        mx = half_max(mx);
        const float msafe = (mx == -INFINITY) ? 0.f : mx;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) sum += __expf(v[j] - msafe);
        sum = half_sum(sum);
        if (l31 == 0 && row < p.M) {
          p.part_m[(size_t)slot * p.M + row] = mx;
          p.part_s[(size_t)slot * p.M + row] = sum;
        }
      }
    return;
  }
  if (p.mode == MODE_RANK) {
    // count, per row, the columns of this wave's sub-tile whose similarity exceeds the row's threshold
    const int slot = tile_n * WN + wn;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float thr = (row < p.M) ? p.lse_row[row] : 0.f;
        const int self = (row < p.M) ? (p.gt ? p.gt[row] : row) : -1;
        float cnt = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int col = n0 + wn * TN + j * 32 + l31;
          cnt += (col < p.N && col != self && acc[i][j][r] * p.alpha > thr) ? 1.f : 0.f;
        }
        cnt = half_sum(cnt);
        if (l31 == 0 && row < p.M) p.part_s[(size_t)slot * p.M + row] = cnt;
      }
    return;
  }
  if (p.mode == MODE_DZ) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * TN + j * 32 + l31;
        const float lc = (col < p.N) ? p.lse_col[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (row < p.M && col < p.N) {
            const float z = acc[i][j][r] * p.alpha;
            float w = __expf(z - p.lse_row[row]) + __expf(z - lc);
            if (col == row + p.offset) w -= 2.0f;
            p.C[(size_t)row * p.ldc + col] = w;
          }
        }
      }
    return;
  }
  // epilogue: acc register r of tile (i,j) is C[row = (r&3) + 8*(r>>2) + 4*half][col = lane&31].
  // All side inputs of a 32x32 tile (residual / aux / old C) are loaded back to back from clamped addresses before
  // any of them is used, so their latency is paid once per tile instead of once per element.
  const int epi = p.slab ? 0 : p.epilogue;
  if (m0 + BM <= p.M && n0 + BN <= p.N) {
    // Interior tile: transpose the accumulators through LDS (the staging buffers are dead: the K loop ended on a
    // barrier) so that every lane moves 16 bytes and a wave-instruction covers whole 512-byte output rows — 4x fewer
    // store (and side-input load) instructions than the register-direct layout, all fully coalesced.
    static_assert(BM * BN <= 2 * (BM + BN) * BK, "C tile must fit in the staging LDS");
    float* ct = lds;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ct[(wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * BN + wn * TN + j * 32 + l31] = acc[i][j][r];
    GEMM_STAMP(11);                    // accumulators written to LDS (includes the wait for the last MFMAs)
    __syncthreads();
    GEMM_STAMP(12);
    // Row stores.  A wave that is not issuing MFMAs gets an instruction through every ~45 cycles while its neighbours
    // on the SIMD are in their K loops (tools/gemm_stamps.py: 9-10 k cycles for this phase with the generic, flag-testing
    // loop), so the epilogue is written for INSTRUCTION COUNT: one buffer descriptor per tensor based at the tile origin,
    // one per-thread byte offset, the chunk step in a scalar offset, the bias loaded once (a thread's chunks share their
    // columns), and one tight loop per epilogue kind instead of flag tests per chunk.
    constexpr int CHUNKS = BM * BN / 4 / NTHR;
    constexpr int RSTEP = NTHR / (BN / 4);                 // rows between a thread's consecutive chunks
    const int lr0 = tid / (BN / 4), lc = (tid % (BN / 4)) * 4;
    const float* ctp = ct + lr0 * BN + lc;
    if (p.slab) {
      float* sl = p.slab + ((size_t)blockIdx.y * p.M + m0 + lr0) * p.N + n0 + lc;
#pragma unroll
      for (int q = 0; q < CHUNKS; ++q)
        *reinterpret_cast<f32x4*>(sl + (size_t)q * RSTEP * p.N) = *reinterpret_cast<const f32x4*>(ctp + q * RSTEP * BN);
      GEMM_STAMP(3);
      GEMM_STAMP_RT(5);
      return;
    }
    const int tile_bytes = (int)(((size_t)(BM - 1) * p.ldc + BN) * 4);
    const size_t org = (size_t)m0 * p.ldc + n0;
    auto tile_rsrc = [&](const float* base) {
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + org), 0, tile_bytes, 0x00020000);
    };
    const int voff = (lr0 * p.ldc + lc) * 4;
    const int sstep = RSTEP * p.ldc * 4;
    auto ld4 = [&](__amdgpu_buffer_rsrc_t r, int q) {
      return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, q * sstep, 0));
    };
    auto st4 = [&](__amdgpu_buffer_rsrc_t r, int q, f32x4 v) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, voff, q * sstep, 0);
    };
    const __amdgpu_buffer_rsrc_t c_rsrc = tile_rsrc(p.C);
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (epi & DCLIP_EPI_BIAS) bv = *reinterpret_cast<const f32x4*>(p.bias + n0 + lc);
    const float alpha = p.alpha;
    const int kind = epi & (DCLIP_EPI_GELU | DCLIP_EPI_DGELU | DCLIP_EPI_RESIDUAL | DCLIP_EPI_ACCUM);
    if (kind == 0) {                                        // [bias]: qkv projection, plain dgrads
#pragma unroll
      for (int q = 0; q < CHUNKS; ++q) st4(c_rsrc, q, *reinterpret_cast<const f32x4*>(ctp + q * RSTEP * BN) * alpha + bv);
    } else if (kind == DCLIP_EPI_RESIDUAL) {                // [bias] + residual: out_proj, fc2, dgrad + skip gradient
      const __amdgpu_buffer_rsrc_t r_rsrc = tile_rsrc(p.residual);
      f32x4 side[CHUNKS];
#pragma unroll
      for (int q = 0; q < CHUNKS; ++q) side[q] = ld4(r_rsrc, q);
#pragma unroll
      for (int q = 0; q < CHUNKS; ++q)
        st4(c_rsrc, q, *reinterpret_cast<const f32x4*>(ctp + q * RSTEP * BN) * alpha + bv + side[q]);
    } else if (kind == DCLIP_EPI_GELU) {                    // [bias] + quick-GELU, pre-activation kept in aux: fc1
      const bool keep = p.aux != nullptr;
      const __amdgpu_buffer_rsrc_t a_rsrc2 = tile_rsrc(keep ? p.aux : p.C);
#pragma unroll
      for (int q = 0; q < CHUNKS; ++q) {
        f32x4 v = *reinterpret_cast<const f32x4*>(ctp + q * RSTEP * BN) * alpha + bv;
        if (keep) st4(a_rsrc2, q, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = quick_gelu_f(v[e]);
        st4(c_rsrc, q, v);
      }
    } else if (kind == DCLIP_EPI_DGELU) {                   // x quick-GELU'(aux): fc2 dgrad
      const __amdgpu_buffer_rsrc_t a_rsrc2 = tile_rsrc(p.aux);
      f32x4 side[CHUNKS];
#pragma unroll
      for (int q = 0; q < CHUNKS; ++q) side[q] = ld4(a_rsrc2, q);
#pragma unroll
      for (int q = 0; q < CHUNKS; ++q) {
        f32x4 v = *reinterpret_cast<const f32x4*>(ctp + q * RSTEP * BN) * alpha + bv;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= quick_gelu_grad_f(side[q][e]);
        st4(c_rsrc, q, v);
      }
    } else {                                                // any other combination: the general form
      const float* s0 = (epi & DCLIP_EPI_DGELU) ? p.aux : p.residual;
      const bool has_s0 = epi & (DCLIP_EPI_RESIDUAL | DCLIP_EPI_DGELU);
#pragma unroll
      for (int q = 0; q < CHUNKS; ++q) {
        const int row = m0 + lr0 + q * RSTEP, col = n0 + lc;
        const size_t off = (size_t)row * p.ldc + col;
        f32x4 v = *reinterpret_cast<const f32x4*>(ctp + q * RSTEP * BN) * alpha + bv;
        f32x4 sd = {0.f, 0.f, 0.f, 0.f};
        if (has_s0) sd = *reinterpret_cast<const f32x4*>(s0 + off);
        if (epi & DCLIP_EPI_GELU) {
          if (p.aux) *reinterpret_cast<f32x4*>(p.aux + off) = v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = quick_gelu_f(v[e]);
        }
        if (epi & DCLIP_EPI_DGELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= quick_gelu_grad_f(sd[e]);
          if (epi & DCLIP_EPI_RESIDUAL) v += *reinterpret_cast<const f32x4*>(p.residual + off);
        } else if (epi & DCLIP_EPI_RESIDUAL) {
          v += sd;
        }
        if (epi & DCLIP_EPI_ACCUM) v += *reinterpret_cast<const f32x4*>(p.C + off);
        *reinterpret_cast<f32x4*>(p.C + off) = v;
      }
    }
    GEMM_STAMP(3);
    GEMM_STAMP_RT(5);
    return;
  }
  const bool has_side0 = epi & (DCLIP_EPI_RESIDUAL | DCLIP_EPI_DGELU);
  const bool has_side1 = epi & DCLIP_EPI_ACCUM;
  float side0[MT][NT][16], side1[MT][NT][16], bcol[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int colc = min(n0 + wn * TN + j * 32 + l31, p.N - 1);
    bcol[j] = (epi & DCLIP_EPI_BIAS) ? p.bias[colc] : 0.f;
  }
  if (has_side0 || has_side1) {
    const float* s0 = (epi & DCLIP_EPI_DGELU) ? p.aux : p.residual;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int colc = min(n0 + wn * TN + j * 32 + l31, p.N - 1);
        const int rbase = m0 + wm * TM + i * 32 + 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const size_t off = (size_t)min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1) * p.ldc + colc;
          side0[i][j][r] = has_side0 ? s0[off] : 0.f;
          side1[i][j][r] = has_side1 ? p.C[off] : 0.f;
        }
      }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + wn * TN + j * 32 + l31;
      const int rbase = m0 + wm * TM + i * 32 + 4 * half;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        if (row < p.M && col < p.N) {
          float v = acc[i][j][r];
          if (p.slab) {
            p.slab[((size_t)blockIdx.y * p.M + row) * p.N + col] = v;
          } else {
            const size_t off = (size_t)row * p.ldc + col;
            v = v * p.alpha + bcol[j];
            if (epi & DCLIP_EPI_GELU) {
              if (p.aux) p.aux[off] = v;
              v = quick_gelu_f(v);
            }
            if (epi & DCLIP_EPI_DGELU) {
              v *= quick_gelu_grad_f(side0[i][j][r]);
              if (epi & DCLIP_EPI_RESIDUAL) v += p.residual[off];
            } else if (epi & DCLIP_EPI_RESIDUAL) {
              v += side0[i][j][r];
            }
            if (has_side1) v += side1[i][j][r];
            p.C[off] = v;
          }
        }
      }
    }
  GEMM_STAMP(3);
  GEMM_STAMP_RT(5);
}

// Sums split-K slabs in fixed order, then applies the epilogue.  One float4 per thread.
__global__ void __launch_bounds__(256) splitk_reduce_kernel(GemmParams p, int splits) {
  const size_t total4 = (size_t)p.M * p.N / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (size_t)gridDim.x * blockDim.x) {
    f32x4 s = *reinterpret_cast<const f32x4*>(p.slab + i * 4);
    for (int z = 1; z < splits; ++z) s += *reinterpret_cast<const f32x4*>(p.slab + (size_t)z * p.M * p.N + i * 4);
    const int row = (int)((i * 4) / p.N), col = (int)((i * 4) % p.N);
#pragma unroll
    for (int e = 0; e < 4; ++e) apply_epilogue_store(p, row, col + e, s[e]);
  }
  if (p.epilogue & DCLIP_EPI_A_ROWSUM) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p.M; i += gridDim.x * blockDim.x) {
      float t = p.rs_slab[i];
      for (int z = 1; z < splits; ++z) t += p.rs_slab[(size_t)z * p.M + i];
      p.aux[i] = t;
    }
  }
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const GemmParams& p_in, int layout, int splits, hipStream_t st) {
  GemmParams p = p_in;
  p.magic_per_group = (unsigned)(0x100000000ull / (unsigned long long)(p.group_m * p.tiles_n));

  dim3 grid(p.tiles_m * p.tiles_n, splits);
  const size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(float);
  const bool ak = layout & DCLIP_A_KMAJOR, bk = layout & DCLIP_B_KMAJOR;
  // LDS-DMA staging for the K-major-A kernels on 128-row tiles (forward and dgrad GEMMs): +1.1 % on the step
  // (3891 vs 3846 img/s, same box).  Measured slower on the 64x64 [K][M]-major wgrad kernel (-1.2 % with everything
  // on DMA), which keeps register staging.  DCLIP_GEMM_DMA=0 switches it off (A/B aid).
  static const bool dma = !(getenv("DCLIP_GEMM_DMA") && atoi(getenv("DCLIP_GEMM_DMA")) == 0);
  if constexpr (BM == 128 && WM * WN == 4) {
    if (dma && ak && p.mode == MODE_GEMM) {
      if (bk) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, true, true>), grid, dim3(WM * WN * 64), lds, st, p);
      else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, false, true>), grid, dim3(WM * WN * 64), lds, st, p);
      return 0;
    }
  }
  if (ak && bk)
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, true>), grid, dim3(WM * WN * 64), lds, st, p);
  else if (ak && !bk)
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, false>), grid, dim3(WM * WN * 64), lds, st, p);
  else if (!ak && bk)
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, true>), grid, dim3(WM * WN * 64), lds, st, p);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, false>), grid, dim3(WM * WN * 64), lds, st, p);
  return 0;
}

constexpr int NUM_CU = 256;

inline int group_m_default() {
  // 4 tile-rows per group: same GEMM time as 8 (58.31 vs 58.29 ms per step, same box) with 10 % less fabric read traffic
  // (387 vs 431 MB per launch; 16: 633 MB and +1 % time) — profiles/r03_l2_hit_rate_and_rasterisation.json
  static const int g = getenv("DCLIP_GEMM_GROUP_M") ? atoi(getenv("DCLIP_GEMM_GROUP_M")) : 4;   // tuning aid
  return g > 0 ? g : 4;
}

struct Plan {
  int bm, bn, splits, k_per_split;
};

// Tile / split-K choice from a small calibrated cost model (MI355X, tools/gemm_tune.py):
//   a CU retires its work items back to back at the MFMA rate whatever the number of co-resident workgroups, so
//   time ~ ceil(items / 256) * (k_tiles * cycles_per_ktile(tile) * eff + fixed(tile)) + slab traffic of a split-K.
// Smaller tiles balance better and have a cheaper prologue/epilogue, larger ones spend fewer LDS reads per MFMA.
Plan make_plan(int M, int N, int K, int layout, int split_k) {
  const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
  if (const char* force = getenv("DCLIP_GEMM_TILE")) {  // tuning aid: "BMxBN[xSPLITS]"
    int bm = 0, bn = 0, sp = 1;
    if (sscanf(force, "%dx%dx%d", &bm, &bn, &sp) >= 2 && (bm == 64 || bm == 128) && (bn == 64 || bn == 128)) {
      if (split_k > 0) sp = split_k;
      int kps = cdiv(cdiv(K, sp), BK) * BK;
      return Plan{bm, bn, cdiv(K, kps), kps};
    }
  }
  // Measured plan for the out-projection's weight gradient of the benched step (768 x 768 x 12,800 tokens): the cost model
  // below only tries power-of-two splits on its preferred tile; 64 x 64 tiles with 7 splits (144 x 7 = 1008 work items = 3.94
  // per CU) measure 122-124 us against 130-131 on two boxes (tools/gemm_wgrad_split_sweep.py; profiles/r03_gemm_wgrad_split_sweep.log).
  // The same sweep's 10 % for the qkv weight gradient did not reproduce on a second box (352-357 us either way) and is not
  // encoded.  DCLIP_GEMM_PLAN_TABLE=0 switches the entry off.
  if (split_k <= 0 && layout == 0 && K == 12800 && M == 768 && N == 768 &&
      !(getenv("DCLIP_GEMM_PLAN_TABLE") && atoi(getenv("DCLIP_GEMM_PLAN_TABLE")) == 0)) {
    const int kps = cdiv(cdiv(K, 7), BK) * BK;
    return Plan{64, 64, cdiv(K, kps), kps};
  }
  const double mn_major_penalty = (layout & DCLIP_A_KMAJOR ? 0.0 : 0.03) + (layout & DCLIP_B_KMAJOR ? 0.0 : 0.03);
  double best = 1e300;
  Plan pl{128, 128, 1, K};
  for (auto& c : cand) {
    const int bm = c[0], bn = c[1];
    const long tiles = (long)cdiv(M, bm) * cdiv(N, bn);
    const double cyc = (bm / 32) * (bn / 32) * 256.0;                       // MFMA cycles per K-tile of this tile
    const double eff = (bm * bn == 16384 ? 1.05 : (bm * bn == 8192 ? 1.06 : 1.10)) + mn_major_penalty;
    const double fixed = bm * bn == 16384 ? 12000.0 : (bm * bn == 8192 ? 7000.0 : 4000.0);
    const int smax = split_k > 0 ? split_k : 32;
    for (int sp = (split_k > 0 ? split_k : 1); sp <= smax; sp *= 2) {
      const int kps = cdiv(cdiv(K, sp), BK) * BK;
      const int s_eff = cdiv(K, kps);
      if (split_k <= 0 && sp > 1 && kps < 512) break;
      const long items = tiles * s_eff;
      double cost = (double)cdiv((int)items, NUM_CU) * ((kps / BK) * cyc * eff + fixed);
      if (s_eff > 1) cost += 8.0 * s_eff * (double)M * N / 1670.0 + 4000.0;  // slab write + read at ~4 TB/s, + a launch
      if (cost < best) {
        best = cost;
        pl = Plan{bm, bn, s_eff, kps};
      }
      if (split_k > 0) break;
    }
  }
  return pl;
}

}  // namespace

#ifdef DCLIP_GEMM_STAMPS
// [workgroups][16] uint64 on the device, or nullptr to stop stamping (diagnostic library only)
DCLIP_API int dclip_debug_set_gemm_stamps(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &buf, sizeof(buf)) == hipSuccess ? DCLIP_OK : DCLIP_ELAUNCH;
}
DCLIP_API void dclip_debug_gemm_plan(int M, int N, int K, int layout, int split_k, int* out4) {
  Plan pl = make_plan(M, N, K, layout, split_k);
  out4[0] = pl.bm, out4[1] = pl.bn, out4[2] = pl.splits, out4[3] = pl.k_per_split;
}
#endif

DCLIP_API size_t dclip_gemm_f32_workspace(int M, int N, int K, int layout, int split_k) {
  Plan pl = make_plan(M, N, K, layout, split_k);
  return pl.splits > 1 ? (size_t)pl.splits * ((size_t)M * N + M) * sizeof(float) : 0;   // C slabs + A_ROWSUM slabs
}

DCLIP_API int dclip_gemm_f32(const float* A, const float* B, float* C, const float* bias,
                             const float* residual, float* aux, int M, int N, int K, int lda, int ldb, int ldc,
                             int layout, int epilogue, float alpha, int split_k, void* workspace,
                             size_t workspace_bytes, void* stream) {
  DCLIP_REQUIRE(A && B && C, "gemm_f32: null operand");
  DCLIP_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_f32: bad shape M=%d N=%d K=%d", M, N, K);
  const bool ak = layout & DCLIP_A_KMAJOR, bk = layout & DCLIP_B_KMAJOR;
  DCLIP_REQUIRE((ak || M % 4 == 0) && lda % 4 == 0 && lda >= (ak ? K : M),
                "gemm_f32: lda (and M when A is [K][M]) must be a multiple of 4 (M=%d K=%d lda=%d)", M, K, lda);
  DCLIP_REQUIRE((bk || N % 4 == 0) && ldb % 4 == 0 && ldb >= (bk ? K : N),
                "gemm_f32: ldb (and N when B is [K][N]) must be a multiple of 4 (N=%d K=%d ldb=%d)", N, K, ldb);
  DCLIP_REQUIRE(N % 4 == 0 && ldc % 4 == 0 && ldc >= N, "gemm_f32: N / ldc must be a multiple of 4");
  DCLIP_REQUIRE(((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) % 16 == 0, "gemm_f32: operands must be 16-byte aligned");
  DCLIP_REQUIRE(!(epilogue & DCLIP_EPI_BIAS) || bias, "gemm_f32: BIAS without bias");
  DCLIP_REQUIRE(!(epilogue & DCLIP_EPI_RESIDUAL) || residual, "gemm_f32: RESIDUAL without residual");
  DCLIP_REQUIRE(!(epilogue & DCLIP_EPI_DGELU) || aux, "gemm_f32: DGELU without aux");
  DCLIP_REQUIRE(!(epilogue & DCLIP_EPI_A_ROWSUM) || (aux && !ak && !(epilogue & (DCLIP_EPI_GELU | DCLIP_EPI_DGELU))),
                "gemm_f32: A_ROWSUM needs aux[M], a [K][M]-major A and no GELU/DGELU (they use aux too)");

  Plan pl = make_plan(M, N, K, layout, split_k);
  // buffer descriptors address 2^31-1 bytes from a work item's origin: a [K][M]-major operand spans k_per_split rows
  DCLIP_REQUIRE(ak || (double)pl.k_per_split * lda * 4.0 < 2147483647.0, "gemm_f32: A slice of %d x %d floats exceeds 2 GiB "
                "per K split; raise split_k", pl.k_per_split, lda);
  DCLIP_REQUIRE(bk || (double)pl.k_per_split * ldb * 4.0 < 2147483647.0, "gemm_f32: B slice of %d x %d floats exceeds 2 GiB "
                "per K split; raise split_k", pl.k_per_split, ldb);
  DCLIP_REQUIRE((double)128 * (ak ? lda : 1) * 4.0 < 2147483647.0 && (double)128 * (bk ? ldb : 1) * 4.0 < 2147483647.0,
                "gemm_f32: leading dimension too large");
  GemmParams p{A, B, C, bias, residual, aux, M, N, K, lda, ldb, ldc, epilogue, alpha,
               cdiv(M, pl.bm), cdiv(N, pl.bn), pl.k_per_split, nullptr, nullptr, group_m_default(),
               MODE_GEMM, nullptr, nullptr, nullptr, nullptr, 0, nullptr};
#ifdef DCLIP_GEMM_STAMPS
  p.dbg = getenv("DCLIP_GEMM_DBG") ? atoi(getenv("DCLIP_GEMM_DBG")) : 0;
#endif
  if (pl.splits > 1) {
    const size_t need = (size_t)pl.splits * ((size_t)M * N + M) * sizeof(float);
    if (!workspace || workspace_bytes < need) {
      dclip_set_error("gemm_f32: split-K needs %zu workspace bytes, got %zu", need, workspace_bytes);
      return DCLIP_EWORKSPACE;
    }
    p.slab = (float*)workspace;
    p.rs_slab = p.slab + (size_t)pl.splits * M * N;
  }
  hipStream_t st = (hipStream_t)stream;
  static const bool w8 = getenv("DCLIP_GEMM_W8") != nullptr;   // experiment: 8-wave 128x128 workgroups
  if (pl.bm == 128 && pl.bn == 128 && w8) launch_cfg<128, 128, 2, 4>(p, layout, pl.splits, st);
  else if (pl.bm == 128 && pl.bn == 128) launch_cfg<128, 128, 2, 2>(p, layout, pl.splits, st);
  else if (pl.bm == 128 && pl.bn == 64) launch_cfg<128, 64, 2, 2>(p, layout, pl.splits, st);
  else if (pl.bm == 64 && pl.bn == 128) launch_cfg<64, 128, 2, 2>(p, layout, pl.splits, st);
  else launch_cfg<64, 64, 2, 2>(p, layout, pl.splits, st);
  DCLIP_CHECK_LAUNCH("gemm_f32");
  if (pl.splits > 1) {
    const size_t total4 = (size_t)M * N / 4;
    int blocks = (int)((total4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p, pl.splits);
    DCLIP_CHECK_LAUNCH("gemm_f32.splitk_reduce");
  }
  return DCLIP_OK;
}


// ------------------------------------------------------------------------------------------------
// Contrastive loss on top of the same MFMA kernel: the [Bl,Bg] logits are produced tile by tile and reduced
// to per-row (max, sum-exp) partials in the epilogue — they are never stored.
namespace {

constexpr int LOSS_BM = 64, LOSS_BN = 64, LOSS_WN = 2;

// one wave per local row: merge the column-tile partials, and compute the positive logit
__global__ void __launch_bounds__(256) lse_merge_kernel(const float* __restrict__ part_m, const float* __restrict__ part_s,
                                                        const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ lse, float* __restrict__ diag, int Bl, int Bg,
                                                        int P, int nslots, int offset, float inv_temp) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Bl) return;
  float m = -INFINITY;
  for (int s = lane; s < nslots; s += 64) m = fmaxf(m, part_m[(size_t)s * Bl + row]);
  m = wave_max(m);
  float sum = 0.f;
  for (int s = lane; s < nslots; s += 64) sum += part_s[(size_t)s * Bl + row] * __expf(part_m[(size_t)s * Bl + row] - m);
  sum = wave_sum(sum);
  float d = 0.f;
  const int col = row + offset;
  if (col >= 0 && col < Bg)
    for (int k = lane; k < P; k += 64) d += a[(size_t)row * P + k] * b[(size_t)col * P + k];
  d = wave_sum(d);
  if (lane == 0) {
    lse[row] = m + __logf(sum);
    diag[row] = d * inv_temp;
  }
}

size_t loss_slots(int Bg) { return (size_t)cdiv(Bg, LOSS_BN) * LOSS_WN; }

}  // namespace

DCLIP_API size_t dclip_contrastive_workspace(int Bl, int Bg, int P) {
  (void)P;
  size_t lse_bytes = 2 * loss_slots(Bg) * (size_t)Bl * sizeof(float);
  size_t dz_bytes = (size_t)Bl * ((Bg + 3) / 4 * 4) * sizeof(float);
  return lse_bytes > dz_bytes ? lse_bytes : dz_bytes;
}

DCLIP_API int dclip_contrastive_lse(const float* a_local, const float* b_global, float* lse, float* diag, int Bl, int Bg,
                                    int P, int offset, float inv_temp, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  DCLIP_REQUIRE(a_local && b_global && lse && diag, "contrastive_lse: null pointer");
  DCLIP_REQUIRE(Bl > 0 && Bg > 0 && P > 0 && P % 4 == 0, "contrastive_lse: bad shape Bl=%d Bg=%d P=%d", Bl, Bg, P);
  const size_t slots = loss_slots(Bg);
  if (!workspace || workspace_bytes < 2 * slots * Bl * sizeof(float)) {
    dclip_set_error("contrastive_lse: workspace too small");
    return DCLIP_EWORKSPACE;
  }
  float* pm = (float*)workspace;
  float* ps = pm + slots * Bl;
  GemmParams p{a_local, b_global, nullptr, nullptr, nullptr, nullptr, Bl, Bg, P, P, P, 0, 0, inv_temp,
               cdiv(Bl, LOSS_BM), cdiv(Bg, LOSS_BN), cdiv(P, BK) * BK, nullptr, nullptr, 8,
               MODE_LSE, nullptr, nullptr, pm, ps, offset, nullptr};
  hipStream_t st = (hipStream_t)stream;
  launch_cfg<LOSS_BM, LOSS_BN, 2, LOSS_WN>(p, DCLIP_A_KMAJOR | DCLIP_B_KMAJOR, 1, st);
  DCLIP_CHECK_LAUNCH("contrastive_lse");
  hipLaunchKernelGGL(lse_merge_kernel, dim3(cdiv(Bl, 4)), dim3(256), 0, st, pm, ps, a_local, b_global, lse, diag, Bl, Bg, P,
                     (int)slots, offset, inv_temp);
  DCLIP_CHECK_LAUNCH("contrastive_lse.merge");
  return DCLIP_OK;
}

DCLIP_API int dclip_contrastive_grad(const float* a_local, const float* b_global, const float* lse_row,
                                     const float* lse_col, float* da_local, int Bl, int Bg, int P, int offset,
                                     float inv_temp, float coef, void* workspace, size_t workspace_bytes, void* stream) {
  DCLIP_REQUIRE(a_local && b_global && lse_row && lse_col && da_local, "contrastive_grad: null pointer");
  DCLIP_REQUIRE(Bl > 0 && Bg > 0 && P > 0 && P % 4 == 0, "contrastive_grad: bad shape");
  const int ldw = (Bg + 3) / 4 * 4;
  if (!workspace || workspace_bytes < (size_t)Bl * ldw * sizeof(float)) {
    dclip_set_error("contrastive_grad: workspace too small");
    return DCLIP_EWORKSPACE;
  }
  float* W = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  // W[i,j] = exp(z_ij - lse_row[i]) + exp(z_ij - lse_col[j]) - 2[j == i+offset]
  GemmParams p{a_local, b_global, W, nullptr, nullptr, nullptr, Bl, Bg, P, P, P, ldw, 0, inv_temp,
               cdiv(Bl, LOSS_BM), cdiv(Bg, LOSS_BN), cdiv(P, BK) * BK, nullptr, nullptr, 8,
               MODE_DZ, lse_row, lse_col, nullptr, nullptr, offset, nullptr};
  launch_cfg<LOSS_BM, LOSS_BN, 2, LOSS_WN>(p, DCLIP_A_KMAJOR | DCLIP_B_KMAJOR, 1, st);
  DCLIP_CHECK_LAUNCH("contrastive_grad.dz");
  // da = coef * W[Bl,Bg] * b_global[Bg,P]   (K = Bg may be ragged; lda = ldw keeps rows 16-byte aligned)
  return dclip_gemm_f32(W, b_global, da_local, nullptr, nullptr, nullptr, Bl, P, Bg, ldw, P, P, DCLIP_A_KMAJOR, 0, coef, 1,
                        nullptr, 0, stream);
}


// ------------------------------------------------------------------------------------------------
// Retrieval / zero-shot ranking on the same similarity tiles (eval_scripts/flickr30k_eval.py:16-88,
// eval_scripts/test_zero_shot_ImageNet.py:82-103): rank of the ground truth = number of candidates scoring higher.
namespace {

__global__ void __launch_bounds__(256) rank_merge_kernel(const float* __restrict__ part, int32_t* __restrict__ count, int Bq,
                                                         int nslots) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= Bq) return;
  float s = 0.f;
  for (int k = 0; k < nslots; ++k) s += part[(size_t)k * Bq + row];
  count[row] = (int32_t)(s + 0.5f);
}

// out[i] = <a_i, b_{idx[i]}>   (one wave per row)
__global__ void __launch_bounds__(256) rowdot_gather_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            const int32_t* __restrict__ idx, float* __restrict__ out, int Bq,
                                                            int Bk, int P) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Bq) return;
  int j = idx ? idx[row] : row;
  j = j < 0 ? 0 : (j >= Bk ? Bk - 1 : j);
  float d = 0.f;
  for (int k = lane; k < P; k += 64) d += a[(size_t)row * P + k] * b[(size_t)j * P + k];
  d = wave_sum(d);
  if (lane == 0) out[row] = d;
}

}  // namespace

DCLIP_API size_t dclip_rank_count_workspace(int Bq, int Bk) { return loss_slots(Bk) * (size_t)Bq * sizeof(float); }

DCLIP_API int dclip_rowdot_gather(const float* a, const float* b, const int32_t* idx, float* out, int Bq, int Bk, int P,
                                  void* stream) {
  DCLIP_REQUIRE(a && b && out && Bq > 0 && Bk > 0 && P > 0, "rowdot_gather: bad arguments");
  hipLaunchKernelGGL(rowdot_gather_kernel, dim3(cdiv(Bq, 4)), dim3(256), 0, (hipStream_t)stream, a, b, idx, out, Bq, Bk, P);
  DCLIP_CHECK_LAUNCH("rowdot_gather");
  return DCLIP_OK;
}

DCLIP_API int dclip_rank_count(const float* queries, const float* candidates, const float* thresh, const int32_t* gt,
                               int32_t* count, int Bq, int Bk, int P, void* workspace, size_t workspace_bytes,
                               void* stream) {
  DCLIP_REQUIRE(queries && candidates && thresh && count, "rank_count: null pointer");
  DCLIP_REQUIRE(Bq > 0 && Bk > 0 && P > 0 && P % 4 == 0, "rank_count: bad shape Bq=%d Bk=%d P=%d", Bq, Bk, P);
  const size_t slots = loss_slots(Bk);
  if (!workspace || workspace_bytes < slots * Bq * sizeof(float)) {
    dclip_set_error("rank_count: workspace too small");
    return DCLIP_EWORKSPACE;
  }
  float* part = (float*)workspace;
  GemmParams p{queries, candidates, nullptr, nullptr, nullptr, nullptr, Bq, Bk, P, P, P, 0, 0, 1.0f,
               cdiv(Bq, LOSS_BM), cdiv(Bk, LOSS_BN), cdiv(P, BK) * BK, nullptr, nullptr, 8,
               MODE_RANK, thresh, nullptr, nullptr, part, 0, gt};
  hipStream_t st = (hipStream_t)stream;
  launch_cfg<LOSS_BM, LOSS_BN, 2, LOSS_WN>(p, DCLIP_A_KMAJOR | DCLIP_B_KMAJOR, 1, st);
  DCLIP_CHECK_LAUNCH("rank_count");
  hipLaunchKernelGGL(rank_merge_kernel, dim3(cdiv(Bq, 256)), dim3(256), 0, st, (const float*)part, count, Bq, (int)slots);
  DCLIP_CHECK_LAUNCH("rank_count.merge");
  return DCLIP_OK;
}
