// fp64 GEMM on the gfx950 matrix cores: C = alpha * op(A) op(B) + beta * C, row-major.
//
// Used for every dense contraction of the sparse-GP path (reference: the XLA:CPU/LAPACK calls
// behind mellon/decomposition.py:115,209 and mellon/conditional.py:57-66,264,522):
//   * the n x m triangular-solve panels   L_j = [L_<j | K_j] W_j^T           (NT)
//   * the Cholesky trailing updates       A22 -= P P^T  (lower tiles only)    (NT)
//   * the Ridge / A A^T Gram              G = L^T L     (split over cells)    (TN)
//   * batched predict                     out = K W                           (NN)
//
// 128x128x16 tile per 256-thread workgroup; 4 waves in a 2x2 grid, each wave owns a 64x64 block
// = 4x4 v_mfma_f64_16x16x4_f64 accumulators (128 VGPRs).  A 64x64x16 variant of the same kernel (2x2
// accumulators per wave) serves launches with fewer tiles than CUs (see launch_dgemm).  Both operand tiles are staged k-major
// in LDS ([k][row], row stride 128+16 doubles so the four k-groups of a wave hit disjoint banks),
// with register prefetch of the next k-tile issued before the MFMA block of the current one.
// fp64 MFMA on gfx950 runs at the fp64 vector rate (64 cycles per 16x16x4 instruction per SIMD),
// so a single LDS buffer + two barriers per k-tile leaves the matrix pipe as the limiter.
#include <cstdlib>
#include <type_traits>
#include "mln_internal.h"
#include "mln_options.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int LPAD = 16;

typedef double d2v __attribute__((ext_vector_type(2)));

// Each thread stages NL doubles of a BT x BKT operand tile (BT = 128 or 64 rows/columns of C).
//   KCONTIG  (element (o, k) at P[o * ld + k]): row o = t / TPR, k range (t % TPR) * NL .. + NL      (TPR = 256 / BT)
//   !KCONTIG (element (o, k) at P[k * ld + o]): k = t / TPK, o range (t % TPK) * NL .. + NL           (TPK = 256 / BKT)
// VEC: 16-byte loads (requires 16-byte aligned base and even ld); element masks still apply.
template <bool KCONTIG, int BT, int BKT, bool VEC>
__device__ __forceinline__ void tile_load(const double* __restrict__ P, int64_t ld, int64_t o0, int64_t k0,
                                          int64_t O, int64_t kend, double (&r)[BT * BKT / 256]) {
  constexpr int NL = BT * BKT / 256;
  const int t = threadIdx.x;
  if (KCONTIG) {
    constexpr int TPR = 256 / BT;
    const int64_t o = o0 + t / TPR;
    const int64_t kb = k0 + (t % TPR) * NL;
    const double* p = P + o * ld + kb;
    const bool ok = o < O;
    if (VEC) {
#pragma unroll
      for (int q = 0; q < NL; q += 2) {
        if (ok && kb + q + 1 < kend) {
          const d2v v = *reinterpret_cast<const d2v*>(p + q);
          r[q] = v.x; r[q + 1] = v.y;
        } else {
          r[q] = (ok && kb + q < kend) ? p[q] : 0.0;
          r[q + 1] = 0.0;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NL; ++q) r[q] = (ok && kb + q < kend) ? p[q] : 0.0;
    }
  } else {
    constexpr int TPK = 256 / BKT;  // threads per k row
    const int64_t k = k0 + t / TPK;
    const int64_t ob = o0 + (t % TPK) * NL;
    const double* p = P + k * ld + ob;
    const bool ok = k < kend;
    if (VEC) {
#pragma unroll
      for (int q = 0; q < NL; q += 2) {
        if (ok && ob + q + 1 < O) {
          const d2v v = *reinterpret_cast<const d2v*>(p + q);
          r[q] = v.x; r[q + 1] = v.y;
        } else {
          r[q] = (ok && ob + q < O) ? p[q] : 0.0;
          r[q + 1] = 0.0;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NL; ++q) r[q] = (ok && ob + q < O) ? p[q] : 0.0;
    }
  }
}

template <bool KCONTIG, int BT, int BKT>
__device__ __forceinline__ void tile_store(double (*S)[BT + LPAD], const double (&r)[BT * BKT / 256]) {
  constexpr int NL = BT * BKT / 256;
  const int t = threadIdx.x;
  if (KCONTIG) {
    constexpr int TPR = 256 / BT;
    const int o = t / TPR, kb = (t % TPR) * NL;
#pragma unroll
    for (int q = 0; q < NL; ++q) S[kb + q][o] = r[q];
  } else {
    constexpr int TPK = 256 / BKT;
    const int k = t / TPK, ob = (t % TPK) * NL;
#pragma unroll
    for (int q = 0; q < NL; q += 2) *reinterpret_cast<d2v*>(&S[k][ob + q]) = (d2v){r[q], r[q + 1]};
  }
}

// BT = 128: the throughput tile (4 x 4 MFMA accumulators per wave).  BT = 64: the latency tile (2 x 2 per wave) for
// launches with few tiles -- the panel and diagonal-block products of the factorisations, where one 128 x 128 x 128
// tile alone is 4.2 MFLOP = 14 us on a single CU at the fp64 matrix rate, and there are 40-80 of them in a chain.
template <bool AK, bool BKC, int BT, int BKT, bool VEC>
__global__ __launch_bounds__(256, 2) void k_dgemm(GemmArgs g, int64_t tiles_n, int64_t kchunk) {
  constexpr int TW = BT / 32;   // MFMA tiles per wave and dimension
  if (g.batch > 1) { g.A += (int64_t)blockIdx.z * g.bsa; g.B += (int64_t)blockIdx.z * g.bsb; g.C += (int64_t)blockIdx.z * g.bsc; }
  __shared__ double As[BKT][BT + LPAD];
  __shared__ double Bs[BKT][BT + LPAD];
  // K-range modes 3 / 7: a tile's work grows with its row block (k <= row), so the row-major launch order ran the heaviest
  // tiles LAST, alone on the chip; reversed, the long ones start first and the short ones fill the tail.
  const bool heavy_last = g.kmode == 3 || g.kmode == 7 || g.kmode == 4;
  const int64_t bid = heavy_last ? ((int64_t)gridDim.x - 1 - blockIdx.x) : (int64_t)blockIdx.x;
  const int64_t tiles_m = ((int64_t)gridDim.x + tiles_n - 1) / tiles_n;
  // (mode 4: k <= column -- the work grows with the COLUMN block: column-major order, reversed)
  const int64_t tm = (g.kmode == 4) ? bid % tiles_m : bid / tiles_n, tn = (g.kmode == 4) ? bid / tiles_m : bid % tiles_n;
  if ((g.lower_only == 1 && tn > tm) || (g.lower_only == 2 && tn >= tm) || (g.lower_only == 3 && tn < tm)) return;
  const int64_t m0 = tm * BT, n0 = tn * BT;
  int64_t kbeg = (int64_t)blockIdx.y * kchunk;
  int64_t kend = (kbeg + kchunk < g.K) ? (kbeg + kchunk) : g.K;
  if (g.kmode == 1) { kbeg = m0; kend = (m0 + BT < g.K) ? (m0 + BT) : g.K; }        // block-diagonal op(A)
  else if (g.kmode == 2) { kbeg = n0; kend = (n0 + BT < g.K) ? (n0 + BT) : g.K; }   // block-diagonal op(B)
  else if (g.kmode == 3) { kend = (m0 + BT < kend) ? (m0 + BT) : kend; }            // op(A) lower triangular: k <= row
  else if (g.kmode == 4) { kend = (n0 + BT < kend) ? (n0 + BT) : kend; }            // op(B) upper triangular: k <= column
  else if (g.kmode == 7) {                                                          // lower x lower: column <= k <= row
    kbeg = (n0 > kbeg) ? n0 : kbeg;
    kend = (m0 + BT < kend) ? (m0 + BT) : kend;
  }
  double* C = g.C + (int64_t)blockIdx.y * g.c_split_stride;  // may alias A (in-place panels)

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * (BT / 2), wn = (wave & 1) * (BT / 2);
  const int lk = lane >> 4, li = lane & 15;

  v4d acc[TW][TW];
#pragma unroll
  for (int i = 0; i < TW; ++i)
#pragma unroll
    for (int j = 0; j < TW; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

  double ra[BT * BKT / 256], rb[BT * BKT / 256];
  if (kbeg < kend) {
    tile_load<AK, BT, BKT, VEC>(g.A, g.lda, m0, kbeg, g.M, kend, ra);
    tile_load<BKC, BT, BKT, VEC>(g.B, g.ldb, n0, kbeg, g.N, kend, rb);
  }
  // beta * C goes INTO the accumulators, requested together with the first operand tiles: read in the epilogue, the C
  // tile was a second exposed memory latency per workgroup (K = 128 updates of the block solves: 31 instead of 42 TFLOP/s)
  const bool use_beta = (g.split_k <= 1) && (g.beta != 0.0);
  const bool beta_in_acc = use_beta && g.alpha != 0.0;
  if (beta_in_acc) {
    const double bs = g.beta / g.alpha;
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
      for (int j = 0; j < TW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = m0 + wm + i * 16 + lk + 4 * r;
          const int64_t col = n0 + wn + j * 16 + li;
          const int64_t rc = (row < g.M) ? row : (g.M - 1), cc = (col < g.N) ? col : (g.N - 1);   // clamped, unconditional
          acc[i][j][r] = bs * C[rc * g.ldc + cc];
        }
  }
  for (int64_t k0 = kbeg; k0 < kend; k0 += BKT) {
    __syncthreads();
    tile_store<AK, BT, BKT>(As, ra);
    tile_store<BKC, BT, BKT>(Bs, rb);
    __syncthreads();
    if (k0 + BKT < kend) {
      tile_load<AK, BT, BKT, VEC>(g.A, g.lda, m0, k0 + BKT, g.M, kend, ra);
      tile_load<BKC, BT, BKT, VEC>(g.B, g.ldb, n0, k0 + BKT, g.N, kend, rb);
    }
#pragma unroll
    for (int kk = 0; kk < BKT; kk += 4) {
      double a[TW], b[TW];
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        a[t] = As[kk + lk][wm + t * 16 + li];
        b[t] = Bs[kk + lk][wn + t * 16 + li];
      }
#pragma unroll
      for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int i = 0; i < TW; ++i)
#pragma unroll
    for (int j = 0; j < TW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm + i * 16 + lk + 4 * r;
        const int64_t col = n0 + wn + j * 16 + li;
        if (row < g.M && col < g.N) {
          double v = g.alpha * acc[i][j][r];
          if (use_beta && !beta_in_acc) v += g.beta * C[row * g.ldc + col];
          C[row * g.ldc + col] = v;
        }
      }
}

// ---- round 4c: the pipelined tile ------------------------------------------------------------------------------------
// The same tile as above with the two things its ISA showed missing: (1) TWO LDS buffers per operand, so that the next
// k-tile's registers are stored while this one's matrix instructions run and ONE barrier per k-tile is left (the single
// buffer needed two, and the 16 ds_write_b64 between them ran with the matrix pipe of that wave idle); (2) the barrier
// waits for LDS traffic only ("s_waitcnt lgkmcnt(0); s_barrier": __syncthreads() also drains vmcnt, i.e. the global loads
// of the tile after next, which are meant to stay in flight across it).  The operand fragments of the next four k are read
// while the 16 (4) matrix instructions of the current four run.
//
// And a launch may MIX the tile sizes: the first `n_big` active tiles (in launch order) are 128 x 128, every later one is
// cut into its four 64 x 64 quadrants, each a workgroup of its own.  722 lower tiles of a Cholesky update or the 1600
// tiles of a 5000 x 5000 product on 512 workgroup slots leave a last round that is 40 % resp. 12 % full; cut in four,
// the stragglers take a quarter of the time each and spread over all CUs.  A quadrant keeps the K range of its PARENT tile
// and every element sums its k in the same order under either tiling: the result is the same bits whatever the mix is.
struct TileMap {
  int64_t tiles_m, tiles_n;   // 128-wide tile grid
  int64_t n_active;           // tiles that are computed (all, or the triangle lower_only names)
  int64_t n_big;              // the first n_big of them (launch order) as 128-tiles, the rest as quadrants
  int order;                  // 0 row-major, 1 row-major reversed, 2 column-major reversed (the heavy-first orders)
  int ring;                   // quadrants with four k-tiles in flight where their K range allows (gemm_tile64_ring)
};

__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// t-th tile of the triangle {tn <= tm} (strict: tn < tm) enumerated row by row
__device__ __forceinline__ void tri_tile(int64_t t, bool strict, int64_t& tm, int64_t& tn) {
  int64_t r = (int64_t)((__dsqrt_rn(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (r * (r + 1) / 2 > t) --r;
  while ((r + 1) * (r + 2) / 2 <= t) ++r;
  tm = strict ? r + 1 : r;
  tn = t - r * (r + 1) / 2;
}

// Unmasked staging loads of a k-tile that lies inside the operand (rows o0 .. o0 + BT below O, k0 + BKT <= kend): `p` is the
// calling thread's element (o, k) = (o0 + its row, k0 + its first k) resp. (k0 + its k, o0 + its first row), see tile_load.
// (Round 4c also measured a branch-free variant of ALL loads -- indices clamped into the operand, zeros selected when the
//  registers go to LDS, with a ring of four register sets for the 64-wide tile: 8192^3 at 0.80 of the fp64 matrix peak
//  against 0.92 for the split below, the 64-wide tile 5-10 % slower as well: the 64-bit clamps cost more issue slots than
//  the masked loads of the few edge tiles cost time.  tools/gemm_mix_probe.py, profiles/r04c_gemm_mix_probe.txt.)
template <bool KCONTIG, int BT, int BKT, bool VEC>
__device__ __forceinline__ void tile_load_inside(const double* __restrict__ p, double (&r)[BT * BKT / 256]) {
  constexpr int NL = BT * BKT / 256;
  if (VEC) {
#pragma unroll
    for (int q = 0; q < NL; q += 2) {
      const d2v v = *reinterpret_cast<const d2v*>(p + q);
      r[q] = v.x; r[q + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int q = 0; q < NL; ++q) r[q] = p[q];
  }
}

template <bool KCONTIG, int BT, int BKT>
__device__ __forceinline__ const double* tile_thread_ptr(const double* P, int64_t ld, int64_t o0, int64_t k0) {
  constexpr int NL = BT * BKT / 256;
  const int t = threadIdx.x;
  if (KCONTIG) {
    constexpr int TPR = 256 / BT;
    return P + (o0 + t / TPR) * ld + k0 + (t % TPR) * NL;
  }
  constexpr int TPK = 256 / BKT;
  return P + (k0 + t / TPK) * ld + o0 + (t % TPK) * NL;
}

template <bool AK, bool BKC, int BT, bool VEC>
__device__ __forceinline__ void gemm_tile_pipelined(const GemmArgs& g, double* C, const int64_t m0, const int64_t n0,
                                                    const int64_t kbeg, const int64_t kend, double* smem) {
  constexpr int BKT = 16;
  constexpr int TW = BT / 32;
  constexpr int LD = BT + LPAD;
  typedef double (*Tile)[LD];
  double* const a_base = smem;                     // As[2][BKT][LD]
  double* const b_base = smem + 2 * BKT * LD;       // Bs[2][BKT][LD]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * (BT / 2), wn = (wave & 1) * (BT / 2);
  const int lk = lane >> 4, li = lane & 15;

  v4d acc[TW][TW];
  double ra[BT * BKT / 256], rb[BT * BKT / 256];
  if (kbeg < kend) {
    tile_load<AK, BT, BKT, VEC>(g.A, g.lda, m0, kbeg, g.M, kend, ra);
    tile_load<BKC, BT, BKT, VEC>(g.B, g.ldb, n0, kbeg, g.N, kend, rb);
  }
  const bool use_beta = (g.split_k <= 1) && (g.beta != 0.0);
  const bool beta_in_acc = use_beta && g.alpha != 0.0;
  if (beta_in_acc) {
    const double bs = g.beta / g.alpha;
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
      for (int j = 0; j < TW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = m0 + wm + i * 16 + lk + 4 * r;
          const int64_t col = n0 + wn + j * 16 + li;
          const int64_t rc = (row < g.M) ? row : (g.M - 1), cc = (col < g.N) ? col : (g.N - 1);
          acc[i][j][r] = bs * C[rc * g.ldc + cc];
        }
  } else {
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
      for (int j = 0; j < TW; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  }
  if (kbeg < kend) {
    tile_store<AK, BT, BKT>(reinterpret_cast<Tile>(a_base), ra);
    tile_store<BKC, BT, BKT>(reinterpret_cast<Tile>(b_base), rb);
    if (kbeg + BKT < kend) {
      tile_load<AK, BT, BKT, VEC>(g.A, g.lda, m0, kbeg + BKT, g.M, kend, ra);
      tile_load<BKC, BT, BKT, VEC>(g.B, g.ldb, n0, kbeg + BKT, g.N, kend, rb);
    }
  }
  lds_only_barrier();
  int cur = 0;
  // this thread's staging addresses for the k-tile two ahead of the one being multiplied (the steady-state loop only)
  const double* pa = tile_thread_ptr<AK, BT, BKT>(g.A, g.lda, m0, kbeg + 2 * BKT);
  const double* pb = tile_thread_ptr<BKC, BT, BKT>(g.B, g.ldb, n0, kbeg + 2 * BKT);
  const int64_t a_step = AK ? (int64_t)BKT : (int64_t)BKT * g.lda, b_step = BKC ? (int64_t)BKT : (int64_t)BKT * g.ldb;

  // one k-tile: fragments of the next four k are read while the matrix instructions of the current four run; after the
  // first group has been issued the registers (k-tile + 1, requested an iteration ago) go into the other LDS buffer --
  // every wave finished reading that one before the barrier that ended the previous iteration -- and k-tile + 2 is requested.
  // STEADY: k-tile + 2 exists and lies inside both operands: no masks, no branches in the body.
  auto k_tile = [&](auto steady, const int64_t k0) {
    constexpr bool STEADY = decltype(steady)::value;
    const Tile Ac = reinterpret_cast<Tile>(a_base + cur * (BKT * LD)), Bc = reinterpret_cast<Tile>(b_base + cur * (BKT * LD));
    const Tile An = reinterpret_cast<Tile>(a_base + (cur ^ 1) * (BKT * LD)), Bn = reinterpret_cast<Tile>(b_base + (cur ^ 1) * (BKT * LD));
    double a[2][TW], b[2][TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      a[0][t] = Ac[lk][wm + t * 16 + li];
      b[0][t] = Bc[lk][wn + t * 16 + li];
    }
#pragma unroll
    for (int kk = 0; kk < BKT; kk += 4) {
      const int s = (kk >> 2) & 1;
      if (kk + 4 < BKT) {
#pragma unroll
        for (int t = 0; t < TW; ++t) {
          a[s ^ 1][t] = Ac[kk + 4 + lk][wm + t * 16 + li];
          b[s ^ 1][t] = Bc[kk + 4 + lk][wn + t * 16 + li];
        }
      }
      if (kk == 4) {
        if (STEADY) {
          tile_store<AK, BT, BKT>(An, ra);
          tile_store<BKC, BT, BKT>(Bn, rb);
          tile_load_inside<AK, BT, BKT, VEC>(pa, ra);
          tile_load_inside<BKC, BT, BKT, VEC>(pb, rb);
          pa += a_step; pb += b_step;
        } else if (k0 + BKT < kend) {
          tile_store<AK, BT, BKT>(An, ra);
          tile_store<BKC, BT, BKT>(Bn, rb);
          if (k0 + 2 * BKT < kend) {
            tile_load<AK, BT, BKT, VEC>(g.A, g.lda, m0, k0 + 2 * BKT, g.M, kend, ra);
            tile_load<BKC, BT, BKT, VEC>(g.B, g.ldb, n0, k0 + 2 * BKT, g.N, kend, rb);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
    }
    lds_only_barrier();
    cur ^= 1;
  };
  int64_t k0 = kbeg;
  if (m0 + BT <= g.M && n0 + BT <= g.N)
    for (; k0 + 3 * BKT <= kend; k0 += BKT) k_tile(std::true_type{}, k0);
  for (; k0 < kend; k0 += BKT) k_tile(std::false_type{}, k0);
#pragma unroll
  for (int i = 0; i < TW; ++i)
#pragma unroll
    for (int j = 0; j < TW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm + i * 16 + lk + 4 * r;
        const int64_t col = n0 + wn + j * 16 + li;
        if (row < g.M && col < g.N) {
          double v = g.alpha * acc[i][j][r];
          if (use_beta && !beta_in_acc) v += g.beta * C[row * g.ldc + col];
          C[row * g.ldc + col] = v;
        }
      }
}

// The 64-wide tile with FOUR k-tiles in flight (a ring of four register sets, 8 registers each).  A chain link of the
// factorisations -- 150 workgroups, K = 128 -- and the quadrants at the end of a mixed launch have their CU to themselves:
// nobody hides the ~2 us a request takes, and with one k-tile in flight each of them waited for it (2.9 us per k-tile of 16
// matrix instructions per wave).  Conditions (the caller checks them): 16-byte loads possible, the K range a whole number
// of GROUPS of four k-tiles (the block sizes of the factorisations are: 128, 256, ...).  Rows past the operand are clamped ONCE, in the thread's base address (they only feed rows / columns of C
// that are never stored), so the requests themselves are unmasked.
template <bool AK, bool BKC>
__device__ __forceinline__ void gemm_tile64_ring(const GemmArgs& g, double* C, const int64_t m0, const int64_t n0,
                                                 const int64_t kbeg, const int64_t kend, double* smem) {
  constexpr int BT = 64, BKT = 16, TW = 2, LD = BT + LPAD, NL = 4, D = 4;
  typedef double (*Tile)[LD];
  double* const a_base = smem;                     // As[2][BKT][LD]
  double* const b_base = smem + 2 * BKT * LD;       // Bs[2][BKT][LD]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = (wave >> 1) * (BT / 2), wn = (wave & 1) * (BT / 2);
  const int lk = lane >> 4, li = lane & 15;
  const int64_t T = (kend - kbeg) / BKT;

  // this thread's two 16-byte pieces of each operand's k-tile (see tile_load for who stages what)
  auto piece_ptrs = [&](auto kcontig, const double* P, int64_t ld, int64_t o0, int64_t O, const double*& q0, const double*& q1, int64_t& step) {
    if (decltype(kcontig)::value) {
      int64_t o = o0 + t / 4;
      o = o < O ? o : O - 1;
      q0 = P + o * ld + kbeg + (t % 4) * NL;
      q1 = q0 + 2;
      step = BKT;
    } else {
      const int64_t omax = (O - 1) & ~(int64_t)1;       // (ld is even: the piece that starts at the last even column is in the row)
      int64_t o = o0 + (t % 16) * NL, o2 = o + 2;
      o = o < omax ? o : omax;
      o2 = o2 < omax ? o2 : omax;
      const double* row = P + (kbeg + t / 16) * ld;
      q0 = row + o; q1 = row + o2;
      step = BKT * ld;
    }
  };
  const double *pa0, *pa1, *pb0, *pb1;
  int64_t a_step, b_step;
  piece_ptrs(std::integral_constant<bool, AK>{}, g.A, g.lda, m0, g.M, pa0, pa1, a_step);
  piece_ptrs(std::integral_constant<bool, BKC>{}, g.B, g.ldb, n0, g.N, pb0, pb1, b_step);
  double ra[D][NL], rb[D][NL];
  int64_t requested = 0;
  // the next k-tile into register set `slot`.  No branch: past the last k-tile the pointers stop and the request is a
  // re-read of that one (never multiplied) -- a branch around the loads makes the compiler's wait-count bookkeeping give up
  // at the join and wait for EVERY outstanding load (vmcnt(0)) before each store, which is the ring undone.
  auto request = [&](auto slot) {
    constexpr int S = decltype(slot)::value;
    const d2v a0 = *reinterpret_cast<const d2v*>(pa0), a1 = *reinterpret_cast<const d2v*>(pa1);
    const d2v b0 = *reinterpret_cast<const d2v*>(pb0), b1 = *reinterpret_cast<const d2v*>(pb1);
    ra[S][0] = a0.x; ra[S][1] = a0.y; ra[S][2] = a1.x; ra[S][3] = a1.y;
    rb[S][0] = b0.x; rb[S][1] = b0.y; rb[S][2] = b1.x; rb[S][3] = b1.y;
    ++requested;
    const int64_t sa = requested < T ? a_step : 0, sb = requested < T ? b_step : 0;
    pa0 += sa; pa1 += sa; pb0 += sb; pb1 += sb;
  };
  request(std::integral_constant<int, 0>{});
  request(std::integral_constant<int, 1>{});
  request(std::integral_constant<int, 2>{});
  request(std::integral_constant<int, 3>{});

  v4d acc[TW][TW];
  const bool use_beta = (g.split_k <= 1) && (g.beta != 0.0);
  const bool beta_in_acc = use_beta && g.alpha != 0.0;
  if (beta_in_acc) {
    const double bs = g.beta / g.alpha;
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
      for (int j = 0; j < TW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = m0 + wm + i * 16 + lk + 4 * r;
          const int64_t col = n0 + wn + j * 16 + li;
          const int64_t rc = (row < g.M) ? row : (g.M - 1), cc = (col < g.N) ? col : (g.N - 1);
          acc[i][j][r] = bs * C[rc * g.ldc + cc];
        }
  } else {
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
      for (int j = 0; j < TW; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
  }
  tile_store<AK, BT, BKT>(reinterpret_cast<Tile>(a_base), ra[0]);
  tile_store<BKC, BT, BKT>(reinterpret_cast<Tile>(b_base), rb[0]);
  request(std::integral_constant<int, 0>{});          // k-tile 4 into the set k-tile 0 has left
  lds_only_barrier();
  int cur = 0;
  // k-tile j is in LDS buffer `cur`, the registers hold k-tiles j + 1 .. j + 4 (k-tile j + 1 in set S1)
  auto k_tile = [&](auto slot1, const int64_t j) {
    constexpr int S1 = decltype(slot1)::value;
    const Tile Ac = reinterpret_cast<Tile>(a_base + cur * (BKT * LD)), Bc = reinterpret_cast<Tile>(b_base + cur * (BKT * LD));
    const Tile An = reinterpret_cast<Tile>(a_base + (cur ^ 1) * (BKT * LD)), Bn = reinterpret_cast<Tile>(b_base + (cur ^ 1) * (BKT * LD));
#pragma unroll
    for (int kk = 0; kk < BKT; kk += 4) {
      double a[TW], b[TW];
#pragma unroll
      for (int tt = 0; tt < TW; ++tt) {
        a[tt] = Ac[kk + lk][wm + tt * 16 + li];
        b[tt] = Bc[kk + lk][wn + tt * 16 + li];
      }
      if (kk == 4) {          // (after the last k-tile: a store nobody reads, a request nobody uses)
        tile_store<AK, BT, BKT>(An, ra[S1]);
        tile_store<BKC, BT, BKT>(Bn, rb[S1]);
        request(std::integral_constant<int, S1>{});
      }
#pragma unroll
      for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j2 = 0; j2 < TW; ++j2)
          acc[i][j2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j2], acc[i][j2], 0, 0, 0);
    }
    lds_only_barrier();
    cur ^= 1;
  };
  for (int64_t j0 = 0; j0 < T; j0 += D) {        // T is a multiple of four (the caller's condition)
    k_tile(std::integral_constant<int, 1>{}, j0);
    k_tile(std::integral_constant<int, 2>{}, j0 + 1);
    k_tile(std::integral_constant<int, 3>{}, j0 + 2);
    k_tile(std::integral_constant<int, 0>{}, j0 + 3);
  }
#pragma unroll
  for (int i = 0; i < TW; ++i)
#pragma unroll
    for (int j = 0; j < TW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm + i * 16 + lk + 4 * r;
        const int64_t col = n0 + wn + j * 16 + li;
        if (row < g.M && col < g.N) {
          double v = g.alpha * acc[i][j][r];
          if (use_beta && !beta_in_acc) v += g.beta * C[row * g.ldc + col];
          C[row * g.ldc + col] = v;
        }
      }
}

// K range of the 128-wide tile at (m0, n0) (the same rules as k_dgemm above)
__device__ __forceinline__ void tile_k_range(const GemmArgs& g, int64_t m0, int64_t n0, int64_t kchunk, int64_t& kbeg, int64_t& kend) {
  kbeg = (int64_t)blockIdx.y * kchunk;
  kend = (kbeg + kchunk < g.K) ? (kbeg + kchunk) : g.K;
  if (g.kmode == 1) { kbeg = m0; kend = (m0 + 128 < g.K) ? (m0 + 128) : g.K; }
  else if (g.kmode == 2) { kbeg = n0; kend = (n0 + 128 < g.K) ? (n0 + 128) : g.K; }
  else if (g.kmode == 3) { kend = (m0 + 128 < kend) ? (m0 + 128) : kend; }
  else if (g.kmode == 4) { kend = (n0 + 128 < kend) ? (n0 + 128) : kend; }
  else if (g.kmode == 7) {
    kbeg = (n0 > kbeg) ? n0 : kbeg;
    kend = (m0 + 128 < kend) ? (m0 + 128) : kend;
  }
}

// MODE 0: 128-tiles and quadrants (252 registers, 72 KB of LDS: two workgroups per CU).  MODE 1: quadrants only, one k-tile in
// flight (95 registers, 40 KB: four per CU) -- launches with more quadrants than two per CU, where the neighbours hide the
// latency and occupancy is what counts (2000^3: 0.35 ms against 0.42 with the ring's 150 registers).  MODE 2: quadrants only,
// four k-tiles in flight (three per CU) -- the chain links.
template <bool AK, bool BKC, bool VEC, int MODE>
__global__ __launch_bounds__(256, MODE == 1 ? 4 : (MODE == 2 ? 3 : 2)) void k_dgemm_mix(GemmArgs g, TileMap tmap, int64_t kchunk) {
  constexpr bool ONLY64 = MODE != 0;
  extern __shared__ __attribute__((aligned(16))) double dgemm_smem[];
  if (g.batch > 1) { g.A += (int64_t)blockIdx.z * g.bsa; g.B += (int64_t)blockIdx.z * g.bsb; g.C += (int64_t)blockIdx.z * g.bsc; }
  const int64_t bid = blockIdx.x;
  const bool big = !ONLY64 && bid < tmap.n_big;
  int64_t t = big ? bid : tmap.n_big + ((bid - tmap.n_big) >> 2);
  const int quad = big ? 0 : (int)((bid - tmap.n_big) & 3);
  if (tmap.order != 0) t = tmap.n_active - 1 - t;
  int64_t tm, tn;
  if (g.lower_only == 1) tri_tile(t, false, tm, tn);
  else if (g.lower_only == 2) tri_tile(t, true, tm, tn);
  else if (g.lower_only == 3) { tri_tile(t, false, tn, tm); }            // upper: the transpose of the lower enumeration
  else if (tmap.order == 2) { tm = t % tmap.tiles_m; tn = t / tmap.tiles_m; }
  else { tm = t / tmap.tiles_n; tn = t % tmap.tiles_n; }
  const int64_t m0 = tm * 128, n0 = tn * 128;
  int64_t kbeg, kend;
  tile_k_range(g, m0, n0, kchunk, kbeg, kend);
  double* C = g.C + (int64_t)blockIdx.y * g.c_split_stride;
  if (!ONLY64 && big) {
    gemm_tile_pipelined<AK, BKC, 128, VEC>(g, C, m0, n0, kbeg, kend, dgemm_smem);
  } else {
    const int64_t qm = m0 + 64 * (quad >> 1), qn = n0 + 64 * (quad & 1);
    if (qm >= g.M || qn >= g.N) return;
    if (VEC && MODE != 1 && tmap.ring && kend - kbeg >= 8 * 16 && ((kend - kbeg) & 63) == 0)
      gemm_tile64_ring<AK, BKC>(g, C, qm, qn, kbeg, kend, dgemm_smem);
    else
      gemm_tile_pipelined<AK, BKC, 64, VEC>(g, C, qm, qn, kbeg, kend, dgemm_smem);
  }
}

__global__ void k_sum_partials(const double* __restrict__ parts, int n_parts, int64_t stride,
                               double* __restrict__ out, int64_t count, double beta) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int p = 0; p < n_parts; ++p) s += parts[(int64_t)p * stride + i];  // fixed order
    out[i] = (beta != 0.0) ? (beta * out[i] + s) : s;
  }
}

}  // namespace

template <int BT, int BKT, bool VEC>
static void dispatch(const GemmArgs& g, dim3 grid, hipStream_t st, int64_t tiles_n, int64_t kchunk) {
  const bool ak = (g.ta == 0), bk = (g.tb == 1);
  dim3 block(256);
  if (ak && bk) hipLaunchKernelGGL((k_dgemm<true, true, BT, BKT, VEC>), grid, block, 0, st, g, tiles_n, kchunk);
  else if (ak && !bk) hipLaunchKernelGGL((k_dgemm<true, false, BT, BKT, VEC>), grid, block, 0, st, g, tiles_n, kchunk);
  else if (!ak && bk) hipLaunchKernelGGL((k_dgemm<false, true, BT, BKT, VEC>), grid, block, 0, st, g, tiles_n, kchunk);
  else hipLaunchKernelGGL((k_dgemm<false, false, BT, BKT, VEC>), grid, block, 0, st, g, tiles_n, kchunk);
}

static int64_t count_active(int lower_only, int64_t tm, int64_t tn) {
  if (lower_only == 0) return tm * tn;
  const int64_t q = tm < tn ? tm : tn;
  if (lower_only == 1) return q * (q + 1) / 2 + (tm > tn ? (tm - tn) * tn : 0);        // tn <= tm
  if (lower_only == 2) return (q > 0 ? q * (q - 1) / 2 : 0) + (tm > tn ? (tm - tn) * tn : 0);   // tn < tm
  return q * (q + 1) / 2 + (tn > tm ? (tn - tm) * tm : 0);                             // tn >= tm
}

template <bool VEC, int MODE>
static hipError_t dispatch_mix(const GemmArgs& g, const TileMap& tmap, dim3 grid, size_t lds, hipStream_t st, int64_t kchunk) {
  const bool ak = (g.ta == 0), bk = (g.tb == 1);
  dim3 block(256);
#define MLN_MIX_LAUNCH(A_, B_)                                                                                              \
  do {                                                                                                                      \
    static bool attr_set = false;                                                                                           \
    if (!attr_set) {                                                                                                        \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_dgemm_mix<A_, B_, VEC, MODE>),                   \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16 * (128 + LPAD) * 8);            \
      if (e != hipSuccess) return e;                                                                                        \
      attr_set = true;                                                                                                      \
    }                                                                                                                       \
    hipLaunchKernelGGL((k_dgemm_mix<A_, B_, VEC, MODE>), grid, block, lds, st, g, tmap, kchunk);                          \
  } while (0)
  if (ak && bk) MLN_MIX_LAUNCH(true, true);
  else if (ak && !bk) MLN_MIX_LAUNCH(true, false);
  else if (!ak && bk) MLN_MIX_LAUNCH(false, true);
  else MLN_MIX_LAUNCH(false, false);
#undef MLN_MIX_LAUNCH
  return hipGetLastError();
}

// MLN_ERR_UNSUPPORTED: not a launch for this kernel (the caller goes on to the single-size kernels)
static int launch_dgemm_mix(mln_ctx* ctx, const GemmArgs& g, int mode, bool any_size) {
  const int64_t tiles_m = (g.M + 127) / 128, tiles_n = (g.N + 127) / 128;
  // the triangle enumeration is written for square tile grids
  if (g.lower_only != 0 && tiles_m != tiles_n) return MLN_ERR_UNSUPPORTED;
  const int64_t n_active = count_active(g.lower_only, tiles_m, tiles_n);
  const int64_t n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
  const int64_t slots = 2 * n_cu;
  const int nbatch = g.batch > 1 ? g.batch : 1;
  const int split = (g.split_k > 1 ? g.split_k : 1) * nbatch;      // (for the tile policy a batch counts like a k-split: more workgroups per tile)
  // (below one round of 128-tiles the launch is all quadrants, served by the quadrant-only instances -- MODE 1 / 2 of
  //  k_dgemm_mix, 40 KB of LDS.  Measured against k_dgemm's 64-wide tiles on the chains of the factorisations (A/B on one box,
  //  tools/r04c_ab.sh): chol(5000) 4.5 -> 4.2 ms, Ridge solve 8.6 -> 8.2 ms, rebuild 17.0 -> 16.5 ms.)
  TileMap tmap;
  tmap.tiles_m = tiles_m; tmap.tiles_n = tiles_n; tmap.n_active = n_active;
  const bool heavy_last = g.kmode == 3 || g.kmode == 7 || g.kmode == 4;
  tmap.order = heavy_last ? ((g.kmode == 4 && g.lower_only == 0) ? 2 : 1) : 0;
  // whole rounds as 128-tiles; a last round that is at least seven eighths full stays 128-wide as well (K = 128 links of the
  // triangular inverses: 384 tiles as 128-tiles 53.6 us, 375 as 1500 quadrants 45.6 us -- the crossover is near 440 of 512)
  const int64_t per_round = slots / split > 0 ? slots / split : 1;
  int64_t n_big = (n_active / per_round) * per_round;
  constexpr int64_t keep8 = 7;      // (tools sweep of round 4c: 6, 7, 8 within 0.3 ms of each other at step level)
  if ((n_active - n_big) * 8 >= keep8 * per_round || mode == 2) n_big = n_active;
  if (mode == 3) n_big = 0;
  tmap.n_big = n_big;
  tmap.ring = 1;      // (the ring for launches of at most two quadrants per CU; "everywhere" measured +0.4 ms: profiles/r04c_ab_ring_everywhere_and_keep8.txt)
  const int64_t nblk = n_big + 4 * (n_active - n_big);
  if (nblk > 0x7fffffffLL) { mln_set_error(ctx, "dgemm grid too large"); return MLN_ERR_UNSUPPORTED; }
  const bool vec = ((uintptr_t)g.A % 16 == 0) && ((uintptr_t)g.B % 16 == 0) && (g.lda % 2 == 0) && (g.ldb % 2 == 0) && (g.bsa % 2 == 0) && (g.bsb % 2 == 0);
  const int ksplit = split / nbatch;
  int64_t kchunk = (g.K + ksplit - 1) / ksplit;
  kchunk = ((kchunk + 15) / 16) * 16;
  if (kchunk <= 0) kchunk = 16;
  dim3 grid((unsigned)nblk, (unsigned)ksplit, (unsigned)nbatch);
  hipError_t e;
  if (n_big == 0) {
    const size_t lds = 4 * 16 * (64 + LPAD) * 8;
    // the ring: launches of at most two quadrants per CU
    if (vec && tmap.ring && nblk * split <= 2 * n_cu) e = dispatch_mix<true, 2>(g, tmap, grid, lds, ctx->stream, kchunk);
    else e = vec ? dispatch_mix<true, 1>(g, tmap, grid, lds, ctx->stream, kchunk) : dispatch_mix<false, 1>(g, tmap, grid, lds, ctx->stream, kchunk);
  } else {
    const size_t lds = 4 * 16 * (128 + LPAD) * 8;
    e = vec ? dispatch_mix<true, 0>(g, tmap, grid, lds, ctx->stream, kchunk) : dispatch_mix<false, 0>(g, tmap, grid, lds, ctx->stream, kchunk);
  }
  MLN_HIP(ctx, e);
  return MLN_OK;
}

static int g_mix_override = -1;  // diagnostics: -1 = policy; 0 = single-size kernels only; 1 = mixed kernel for every launch it can serve
void dgemm_set_mix(int mode) { g_mix_override = mode; }

int launch_dgemm(mln_ctx* ctx, const GemmArgs& g) {
  if (g.M <= 0 || g.N <= 0) return MLN_OK;
  // tile choice: 128 x 128 unless that leaves most CUs idle (fewer tiles than CUs): then 64 x 64 -- four times the
  // workgroups, a quarter of the serial work each.  In-place panel products (C aliases A with one column tile per row
  // tile) keep 128 when N > 64, because two 64-wide column tiles of one row tile would read what the other writes.
  // (callers that update in place were written for 128-wide tiles: any overlap of C with an operand keeps them)
  auto overlaps = [&](const double* P, int64_t rows, int64_t cols, int64_t ld) {
    const double* c1 = g.C + (g.M - 1) * g.ldc + g.N;
    const double* p1 = P + (rows - 1) * ld + cols;
    if (!(g.C < p1 && P < c1)) return false;            // disjoint address ranges
    if (ld != g.ldc) return true;                        // different layouts: assume the worst
    // two rectangles of one row-major matrix with leading dimension ld: compare row and column intervals
    const int64_t off = P - g.C;                         // position of P relative to C
    int64_t dr = off / ld, dc = off % ld;
    if (dc < 0) { dc += ld; dr -= 1; }
    if (dc + cols > ld) return true;                     // wraps around a row end: not a rectangle of this matrix
    const bool rows_hit = dr < g.M && 0 < dr + rows;
    const bool cols_hit = dc < g.N && 0 < dc + cols;
    // (P to the left of C in the same rows shows up as dc close to ld with dr one less)
    const bool cols_hit_wrapped = (dc - ld) < g.N && 0 < (dc - ld) + cols && (dr + 1) < g.M && 0 < dr + 1 + rows;
    return (rows_hit && cols_hit) || cols_hit_wrapped;
  };
  const bool inplace = overlaps(g.A, g.ta == 0 ? g.M : g.K, g.ta == 0 ? g.K : g.M, g.lda) ||
                       overlaps(g.B, g.tb == 0 ? g.K : g.N, g.tb == 0 ? g.N : g.K, g.ldb);
  int bt = 128;
  {
    const int64_t t128 = ((g.M + 127) / 128) * ((g.N + 127) / 128) / (g.lower_only ? 2 : 1);
    const int64_t n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
    // (below 4 tiles per CU the 128-wide tiling leaves its last round of workgroups mostly empty -- 722 tiles on 512 slots
    //  -- and four times as many 64-wide tiles pack better: factor + inverses 9.1 -> 8.5 ms, rebuild 17.7 -> 17.1 ms at m = 5000)
    if (t128 * (g.split_k > 1 ? g.split_k : 1) < 4 * n_cu && !inplace) bt = 64;
  }
  if (g.kmode == 1 || g.kmode == 2) bt = 128;   // the block-diagonal modes are defined on 128-wide blocks
  {
    // the pipelined kernel with mixed tile sizes (see TileMap): whole rounds of 128-tiles, the rest as quadrants
    constexpr int mix_mode = 1;
    if (g_mix_override != 0 && mix_mode > 0 && !inplace) {
      int rc = launch_dgemm_mix(ctx, g, mix_mode, g_mix_override == 1);
      if (rc != MLN_ERR_UNSUPPORTED) return rc;
    }
  }
  const int64_t tiles_m = (g.M + bt - 1) / bt, tiles_n = (g.N + bt - 1) / bt;
  const int64_t nblk = tiles_m * tiles_n;
  if (nblk > 0x7fffffffLL) { mln_set_error(ctx, "dgemm grid too large"); return MLN_ERR_UNSUPPORTED; }
  const bool vec = ((uintptr_t)g.A % 16 == 0) && ((uintptr_t)g.B % 16 == 0) && (g.lda % 2 == 0) && (g.ldb % 2 == 0) && (g.bsa % 2 == 0) && (g.bsb % 2 == 0);
  constexpr int bkt = 16;   // measured: BK=32 (246 VGPRs, 74 KB LDS) is 10-20 % slower than BK=16 on every shape
  int split = g.split_k > 1 ? g.split_k : 1;
  int64_t kchunk = (g.K + split - 1) / split;
  kchunk = ((kchunk + bkt - 1) / bkt) * bkt;
  if (kchunk <= 0) kchunk = bkt;
  dim3 grid((unsigned)nblk, (unsigned)split, (unsigned)(g.batch > 1 ? g.batch : 1));
  if (bt == 64) {
    if (vec) dispatch<64, 16, true>(g, grid, ctx->stream, tiles_n, kchunk);
    else dispatch<64, 16, false>(g, grid, ctx->stream, tiles_n, kchunk);
  } else {
    if (vec) dispatch<128, 16, true>(g, grid, ctx->stream, tiles_n, kchunk);
    else dispatch<128, 16, false>(g, grid, ctx->stream, tiles_n, kchunk);
  }
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_sum_partials(mln_ctx* ctx, const double* parts, int n_parts, int64_t stride, double* out,
                        int64_t count, double beta) {
  if (count <= 0) return MLN_OK;
  int64_t nb = (count + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)nb), dim3(256), 0, ctx->stream, parts, n_parts, stride, out,
                     count, beta);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}
