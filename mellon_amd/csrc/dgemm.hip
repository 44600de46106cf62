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
// = 4x4 v_mfma_f64_16x16x4_f64 accumulators (128 VGPRs).  Both operand tiles are staged k-major
// in LDS ([k][row], row stride 128+16 doubles so the four k-groups of a wave hit disjoint banks),
// with register prefetch of the next k-tile issued before the MFMA block of the current one.
// fp64 MFMA on gfx950 runs at the fp64 vector rate (64 cycles per 16x16x4 instruction per SIMD),
// so a single LDS buffer + two barriers per k-tile leaves the matrix pipe as the limiter.
#include "mln_internal.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, LPAD = 16;

typedef double d2v __attribute__((ext_vector_type(2)));

// Each thread stages NL = BKT/2 doubles of a 128 x BKT operand tile.
//   KCONTIG  (element (o, k) at P[o * ld + k]): row o = t / 2, k range (t % 2) * NL .. + NL
//   !KCONTIG (element (o, k) at P[k * ld + o]): k = t / (256 / BKT), o range (t % (256/BKT)) * NL .. + NL
// VEC: 16-byte loads (requires 16-byte aligned base and even ld); element masks still apply.
template <bool KCONTIG, int BKT, bool VEC>
__device__ __forceinline__ void tile_load(const double* __restrict__ P, int64_t ld, int64_t o0, int64_t k0,
                                          int64_t O, int64_t kend, double (&r)[BKT / 2]) {
  constexpr int NL = BKT / 2;
  const int t = threadIdx.x;
  if (KCONTIG) {
    const int64_t o = o0 + (t >> 1);
    const int64_t kb = k0 + (t & 1) * NL;
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

template <bool KCONTIG, int BKT>
__device__ __forceinline__ void tile_store(double (*S)[BM + LPAD], const double (&r)[BKT / 2]) {
  constexpr int NL = BKT / 2;
  const int t = threadIdx.x;
  if (KCONTIG) {
    const int o = t >> 1, kb = (t & 1) * NL;
#pragma unroll
    for (int q = 0; q < NL; ++q) S[kb + q][o] = r[q];
  } else {
    constexpr int TPK = 256 / BKT;
    const int k = t / TPK, ob = (t % TPK) * NL;
#pragma unroll
    for (int q = 0; q < NL; q += 2) *reinterpret_cast<d2v*>(&S[k][ob + q]) = (d2v){r[q], r[q + 1]};
  }
}

template <bool AK, bool BKC, int BKT, bool VEC>
__global__ __launch_bounds__(256, 2) void k_dgemm(GemmArgs g, int64_t tiles_n, int64_t kchunk) {
  __shared__ double As[BKT][BM + LPAD];
  __shared__ double Bs[BKT][BN + LPAD];
  const int64_t bid = blockIdx.x;
  const int64_t tm = bid / tiles_n, tn = bid % tiles_n;
  if (g.lower_only && tn > tm) return;
  const int64_t m0 = tm * BM, n0 = tn * BN;
  const int64_t kbeg = (int64_t)blockIdx.y * kchunk;
  const int64_t kend = (kbeg + kchunk < g.K) ? (kbeg + kchunk) : g.K;
  double* C = g.C + (int64_t)blockIdx.y * g.c_split_stride;  // may alias A (in-place panels)

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int lk = lane >> 4, li = lane & 15;

  v4d acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

  double ra[BKT / 2], rb[BKT / 2];
  if (kbeg < kend) {
    tile_load<AK, BKT, VEC>(g.A, g.lda, m0, kbeg, g.M, kend, ra);
    tile_load<BKC, BKT, VEC>(g.B, g.ldb, n0, kbeg, g.N, kend, rb);
  }
  for (int64_t k0 = kbeg; k0 < kend; k0 += BKT) {
    __syncthreads();
    tile_store<AK, BKT>(As, ra);
    tile_store<BKC, BKT>(Bs, rb);
    __syncthreads();
    if (k0 + BKT < kend) {
      tile_load<AK, BKT, VEC>(g.A, g.lda, m0, k0 + BKT, g.M, kend, ra);
      tile_load<BKC, BKT, VEC>(g.B, g.ldb, n0, k0 + BKT, g.N, kend, rb);
    }
#pragma unroll
    for (int kk = 0; kk < BKT; kk += 4) {
      double a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = As[kk + lk][wm + t * 16 + li];
        b[t] = Bs[kk + lk][wn + t * 16 + li];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  const bool use_beta = (g.split_k <= 1) && (g.beta != 0.0);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm + i * 16 + lk + 4 * r;
        const int64_t col = n0 + wn + j * 16 + li;
        if (row < g.M && col < g.N) {
          double v = g.alpha * acc[i][j][r];
          if (use_beta) v += g.beta * C[row * g.ldc + col];
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

template <int BKT, bool VEC>
static void dispatch(const GemmArgs& g, dim3 grid, hipStream_t st, int64_t tiles_n, int64_t kchunk) {
  const bool ak = (g.ta == 0), bk = (g.tb == 1);
  dim3 block(256);
  if (ak && bk) hipLaunchKernelGGL((k_dgemm<true, true, BKT, VEC>), grid, block, 0, st, g, tiles_n, kchunk);
  else if (ak && !bk) hipLaunchKernelGGL((k_dgemm<true, false, BKT, VEC>), grid, block, 0, st, g, tiles_n, kchunk);
  else if (!ak && bk) hipLaunchKernelGGL((k_dgemm<false, true, BKT, VEC>), grid, block, 0, st, g, tiles_n, kchunk);
  else hipLaunchKernelGGL((k_dgemm<false, false, BKT, VEC>), grid, block, 0, st, g, tiles_n, kchunk);
}

static int g_bk_override = -1;   // diagnostics: force BK (16 / 32); -1 = automatic
void dgemm_set_bk(int bk) { g_bk_override = bk; }

int launch_dgemm(mln_ctx* ctx, const GemmArgs& g) {
  if (g.M <= 0 || g.N <= 0) return MLN_OK;
  const int64_t tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  const int64_t nblk = tiles_m * tiles_n;
  if (nblk > 0x7fffffffLL) { mln_set_error(ctx, "dgemm grid too large"); return MLN_ERR_UNSUPPORTED; }
  const bool vec = ((uintptr_t)g.A % 16 == 0) && ((uintptr_t)g.B % 16 == 0) && (g.lda % 2 == 0) && (g.ldb % 2 == 0);
  int bkt = 16;   // measured: BK=32 (246 VGPRs, 74 KB LDS) is 10-20 % slower than BK=16 on every shape
  if (g_bk_override == 16 || g_bk_override == 32) bkt = g_bk_override;
  int split = g.split_k > 1 ? g.split_k : 1;
  int64_t kchunk = (g.K + split - 1) / split;
  kchunk = ((kchunk + bkt - 1) / bkt) * bkt;
  if (kchunk <= 0) kchunk = bkt;
  dim3 grid((unsigned)nblk, (unsigned)split);
  if (bkt == 32) {
    if (vec) dispatch<32, true>(g, grid, ctx->stream, tiles_n, kchunk);
    else dispatch<32, false>(g, grid, ctx->stream, tiles_n, kchunk);
  } else {
    if (vec) dispatch<16, true>(g, grid, ctx->stream, tiles_n, kchunk);
    else dispatch<16, false>(g, grid, ctx->stream, tiles_n, kchunk);
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
