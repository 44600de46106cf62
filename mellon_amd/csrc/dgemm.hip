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

static int g_bk_override = -1;   // diagnostics: force the tile size (128 / 64); -1 = automatic
void dgemm_set_bk(int bk) { g_bk_override = bk; }

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
    static const int64_t below = mln_experiment("MELLON_AMD_GEMM64_BELOW") ? std::atoll(mln_experiment("MELLON_AMD_GEMM64_BELOW")) : 0;
    // (below 4 tiles per CU the 128-wide tiling leaves its last round of workgroups mostly empty -- 722 tiles on 512 slots
    //  -- and four times as many 64-wide tiles pack better: factor + inverses 9.1 -> 8.5 ms, rebuild 17.7 -> 17.1 ms at m = 5000)
    if (t128 * (g.split_k > 1 ? g.split_k : 1) < (below > 0 ? below : 4 * n_cu) && !inplace) bt = 64;
  }
  if (g_bk_override == 128 || g_bk_override == 64) bt = inplace ? 128 : g_bk_override;
  if (g.kmode == 1 || g.kmode == 2) bt = 128;   // the block-diagonal modes are defined on 128-wide blocks
  const int64_t tiles_m = (g.M + bt - 1) / bt, tiles_n = (g.N + bt - 1) / bt;
  const int64_t nblk = tiles_m * tiles_n;
  if (nblk > 0x7fffffffLL) { mln_set_error(ctx, "dgemm grid too large"); return MLN_ERR_UNSUPPORTED; }
  const bool vec = ((uintptr_t)g.A % 16 == 0) && ((uintptr_t)g.B % 16 == 0) && (g.lda % 2 == 0) && (g.ldb % 2 == 0);
  constexpr int bkt = 16;   // measured: BK=32 (246 VGPRs, 74 KB LDS) is 10-20 % slower than BK=16 on every shape
  int split = g.split_k > 1 ? g.split_k : 1;
  int64_t kchunk = (g.K + split - 1) / split;
  kchunk = ((kchunk + bkt - 1) / bkt) * bkt;
  if (kchunk <= 0) kchunk = bkt;
  dim3 grid((unsigned)nblk, (unsigned)split);
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
