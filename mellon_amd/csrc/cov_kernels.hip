// Pairwise-distance + covariance tiles (reference: mellon/util.py:351-366 `distance`,
// mellon/cov.py k() of Matern32/52, ExpQuad, Exponential, RatQuad, Linear and
// mellon/base_cov.py:301-453 Add/Mul/Pow), fp64, gfx950.
//
// Layout: 64 cells x 64 centres per 256-thread workgroup, 4x4 outputs per thread.  The cell and
// centre tiles are staged k-major in LDS in chunks of 16 active dims (coalesced 128-B row reads),
// the x.y dot product is an fp64 FMA chain in registers, and the sqrt/exp epilogue runs on the
// same registers -- the n x m distance matrix never exists in HBM.  d <= 51 is too skinny for
// the fp64 MFMA (same rate as v_fma_f64 on gfx950) to pay, so this is a VALU kernel.
#include <cstdlib>
#include <vector>

#include "mln_internal.h"
#include "rowmin_f16.h"
#include "predict_rows_prod.h"
#include "cov_program.h"
#include "cov_rows.h"
#include "cov_rows_q.h"
#include "mln_options.h"

namespace {

constexpr int TM = 64, TN = covrows::TN, DK = 16, PADT = 4;
constexpr int NNS = covrows::NNS;

// Divisions by the length scale / by 3 are multiplications by the reciprocal (one rounding instead of
// a ~20-instruction fp64 divide per element): values differ from the reference's `x / ls` by <= 1 ulp.
__device__ __forceinline__ double leaf_value(const DevLeaf& lf, double xx, double yy, double xy) {
  const double inv_ls = lf.alpha_inv_ls[1];
  if (lf.kind == MLN_K_LINEAR) return xy * inv_ls;          // cov.py:554
  // util.py:362-366: sq = xx - 2 xy + yy + 1e-12 ; dist = sqrt(max(sq, 0))
  double sq = xx - 2.0 * xy + yy + 1e-12;
  double dist = sqrt(fmax(sq, 0.0));
  switch (lf.kind) {
    case MLN_K_DISTANCE: return dist * inv_ls;              // util.py:366
    case MLN_K_MATERN32: {                                  // cov.py:64-65
      double r = 1.7320508075688772 * dist * inv_ls;
      return (r + 1.0) * exp(-r);
    }
    case MLN_K_MATERN52: {                                  // cov.py:159-160
      double r = 2.23606797749979 * dist * inv_ls;
      return (r + r * r * 0.3333333333333333 + 1.0) * exp(-r);
    }
    case MLN_K_EXPQUAD: {                                   // cov.py:257-258
      double r = dist * inv_ls;
      return exp(-0.5 * (r * r));
    }
    case MLN_K_EXPONENTIAL: {                               // cov.py:354-355
      double r = dist * inv_ls;
      return exp(-0.5 * r);
    }
    default: {                                              // RatQuad cov.py:455-456
      double r = dist * inv_ls;
      return pow(r * r / (2.0 * lf.alpha) + 1.0, -lf.alpha);
    }
  }
}

// Radial factor g of d leaf(x, y)/dx = g * (x - y)|dims (stationary leaves), the exact derivative of what
// leaf_value evaluates (denominator dist, 0 where the clamp of util.distance is active): the formulas of
// cov.py k_grad (cov.py:84-97,186-199,283-296,380-393,481-496) with util.distance_grad's +1e-12 dropped.
__device__ __forceinline__ double leaf_grad_coeff(const DevLeaf& lf, double xx, double yy, double xy) {
  const double inv_ls = lf.alpha_inv_ls[1];
  const double sq = xx - 2.0 * xy + yy + 1e-12;
  const double dist = sqrt(fmax(sq, 0.0));
  const double inv = (sq > 0.0) ? 1.0 / dist : 0.0;
  switch (lf.kind) {
    case MLN_K_MATERN32: {
      const double f = 1.7320508075688772 * inv_ls, r = f * dist;
      return -f * r * exp(-r) * inv;
    }
    case MLN_K_MATERN52: {
      const double f = 2.23606797749979 * inv_ls, r = f * dist;
      return -0.3333333333333333 * exp(-r) * r * (r + 1.0) * f * inv;
    }
    case MLN_K_EXPQUAD: {
      const double r = dist * inv_ls;
      return -r * inv_ls * exp(-0.5 * (r * r)) * inv;
    }
    case MLN_K_EXPONENTIAL: {
      const double r = dist * inv_ls;
      return -0.5 * inv_ls * exp(-0.5 * r) * inv;
    }
    default: {
      const double r = dist * inv_ls, b = r * r / (2.0 * lf.alpha) + 1.0;
      return -r * inv_ls * pow(b, -lf.alpha - 1.0) * inv;
    }
  }
}

// acc[i][j] = sum_k x[row0+ty*4+i][dims[k]] * y[col0+tx*4+j][dims[k]] over one leaf's active dims
__device__ __forceinline__ void leaf_dot(const DevCov& cov, const DevLeaf& lf, const double* __restrict__ x,
                                         int64_t n, const double* __restrict__ y, int64_t m, int d,
                                         int64_t row0, int64_t col0, double (*xs)[TM + PADT],
                                         double (*ys)[TN + PADT], double acc[4][4]) {
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  for (int k0 = 0; k0 < lf.ndims; k0 += DK) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int idx = tid + 256 * q;
      int r = idx >> 4, k = idx & 15;
      double vx = 0.0, vy = 0.0;
      if (k0 + k < lf.ndims) {
        int col = cov.dims[lf.dims_off + k0 + k];
        if (row0 + r < n) vx = x[(row0 + r) * (int64_t)d + col];
        if (col0 + r < m) vy = y[(col0 + r) * (int64_t)d + col];
      }
      xs[k][r] = vx;
      ys[k][r] = vy;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < DK; ++k) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = xs[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = ys[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
    }
  }
}

// Evaluates the whole covariance program for this thread's 4x4 outputs into val.
template <bool SINGLE, bool GRADC = false>
__device__ __forceinline__ void cov_tile(const DevCov& cov, const double* __restrict__ x, int64_t n,
                                         const double* __restrict__ y, int64_t m, int d,
                                         const double* __restrict__ xx, const double* __restrict__ yy,
                                         int64_t row0, int64_t col0, double (*xs)[TM + PADT],
                                         double (*ys)[TN + PADT], double val[4][4]) {
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double s1[4][4], s2[4][4];  // stack below the top (val is the top); max depth 3
  int sp = 0;
  const int ntok = SINGLE ? 1 : cov.n_toks;
  for (int t = 0; t < ntok; ++t) {
    const int op = SINGLE ? MLN_OP_LEAF : cov.tok_op[t];
    if (op == MLN_OP_LEAF || op == MLN_OP_CONST) {
      if (!SINGLE && sp > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) { s2[i][j] = s1[i][j]; s1[i][j] = val[i][j]; }
      }
      if (op == MLN_OP_CONST) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) val[i][j] = cov.tok_val[t];
      } else {
        const int li = SINGLE ? 0 : cov.tok_leaf[t];
        const DevLeaf lf = cov.leaves[li];
        double acc[4][4];
        leaf_dot(cov, lf, x, n, y, m, d, row0, col0, xs, ys, acc);
        double xr[4], yr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int64_t r = row0 + ty * 4 + i;
          xr[i] = (r < n) ? xx[(int64_t)li * n + r] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int64_t c = col0 + tx * 4 + j;
          yr[j] = (c < m) ? yy[(int64_t)li * m + c] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            val[i][j] = GRADC ? leaf_grad_coeff(lf, xr[i], yr[j], acc[i][j]) : leaf_value(lf, xr[i], yr[j], acc[i][j]);
      }
      ++sp;
    } else {
      // binary op: left = s1, right = val (top)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          double l = s1[i][j], r = val[i][j];
          val[i][j] = (op == MLN_OP_ADD) ? (l + r) : (op == MLN_OP_MUL) ? (l * r) : pow(l, r);
          s1[i][j] = s2[i][j];
        }
      --sp;
    }
  }
}

template <bool SINGLE>
__global__ __launch_bounds__(256) void k_kernel_matrix(DevCov cov, const double* __restrict__ x, int64_t n,
                                                       const double* __restrict__ y, int64_t m, int d,
                                                       const double* __restrict__ xx,
                                                       const double* __restrict__ yy,
                                                       double* __restrict__ out, int64_t ldo,
                                                       double add_diag, int64_t tiles_n,
                                                       float* __restrict__ out32, int q32) {
  __shared__ double xs[DK][TM + PADT];
  __shared__ double ys[DK][TN + PADT];
  const int64_t bid = blockIdx.x;
  const int64_t row0 = (bid / tiles_n) * TM, col0 = (bid % tiles_n) * TN;
  double val[4][4];
  cov_tile<SINGLE>(cov, x, n, y, m, d, xx, yy, row0, col0, xs, ys, val);
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int64_t r = row0 + ty * 4 + i;
    if (r >= n) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int64_t c = col0 + tx * 4 + j;
      const double v = (c < m) ? val[i][j] + ((r == c) ? add_diag : 0.0) : 0.0;
      if (c < ldo) {   // pad columns of the leading dimension stay zero
        out[r * ldo + c] = v;
        if (out32) out32[r * ldo + c] = mln_surrogate_bits(v, q32);   // 32-bit copy for the warm-up passes of the MAP solve
      }
    }
  }
}

// Single-leaf kernel matrix with the x.y dot products on the matrix cores: for one leaf over d contiguous
// columns the dot product is a plain (64 x d) x (d x 64) GEMM tile.  v_mfma_f64_16x16x4 with both operands
// read straight from global memory / L1 in the instruction's own layout (A[m = li][k = lk], B[k = lk][n = li];
// a tile of x or y is <= 26 KB and is re-read from L1 by the four waves).  Wave w owns rows 16 w .. 16 w + 15
// of the 64 x 64 tile (4 accumulators); the sqrt/exp epilogue and the fp64 (+ optional fp32) stores run on
// the accumulator layout D[row = lk + 4 reg][col = li].  Measured at 1e6 x 5000 x 50: 32 ms against 35 ms for
// the LDS-tiled VALU kernel above (PMC: VALU ~50 % busy on ~130 instructions per element, matrix pipe 20 %);
// staging this one-tile-per-workgroup form through LDS (38 ms) and hand-written sqrt/exp (no change) were
// tried and dropped; large shapes use k_kernel_matrix_rows below, which overlaps the matrix pipe with the
// epilogue inside each wave (27 ms).

__global__ __launch_bounds__(256) void k_kernel_matrix_mfma(DevCov cov, const double* __restrict__ x, int64_t n,
                                                            const double* __restrict__ y, int64_t m, int d,
                                                            const double* __restrict__ xx,
                                                            const double* __restrict__ yy,
                                                            double* __restrict__ out, int64_t ldo, double add_diag,
                                                            int64_t tiles_n, float* __restrict__ out32, int q32) {
  const DevLeaf lf = cov.leaves[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
  const int64_t bid = blockIdx.x;
  const int64_t row0 = (bid / tiles_n) * TM + wave * 16, col0 = (bid % tiles_n) * TN;
  // operand rows (clamped: loads stay unconditional, invalid rows / columns are never stored)
  const int64_t ar = (row0 + li < n) ? row0 + li : n - 1;
  const double* __restrict__ xa = x + ar * d;
  const double* yb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int64_t c = col0 + 16 * t + li;
    yb[t] = y + ((c < m) ? c : m - 1) * d;
  }
  v4d_t acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = v4d_t{0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < d; k0 += 4) {
    const int k = k0 + lk;
    const int kc = (k < d) ? k : d - 1;
    const double mask = (k < d) ? 1.0 : 0.0;
    const double a = xa[kc] * mask;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, yb[t][kc], acc[t], 0, 0, 0);
  }
  double xr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = row0 + lk + 4 * r;
    xr[r] = (row < n) ? xx[row] : 0.0;
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int64_t c = col0 + 16 * t + li;
    const double yc = (c < m) ? yy[c] : 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = row0 + lk + 4 * r;
      if (row < n && c < ldo) {   // pad columns of the leading dimension stay zero
        const double v = (c < m) ? leaf_value(lf, xr[r], yc, acc[t][r]) + ((row == c) ? add_diag : 0.0) : 0.0;
        out[row * ldo + c] = v;
        if (out32) out32[row * ldo + c] = mln_surrogate_bits(v, q32);
      }
    }
  }
}

// ---- predictor gradient, single stationary leaf: grad_i = sum_j w_j g_ij (x_i - c_j)|dims -------------------
//   Q (rows x m) = w_j g_ij                     one covariance-tile pass (this kernel)
//   T = Q [C|dims , 1]                          one GEMM on the matrix cores
//   grad_i[dims] = x_i[dims] T_i,last - T_i,k   (k_grad_combine)
__global__ __launch_bounds__(256) void k_grad_coeff(DevCov cov, const double* __restrict__ x, int64_t n,
                                                    const double* __restrict__ y, int64_t m, int d,
                                                    const double* __restrict__ xx, const double* __restrict__ yy,
                                                    const double* __restrict__ w, double* __restrict__ out,
                                                    int64_t ldo, int64_t tiles_n) {
  __shared__ double xs[DK][TM + PADT];
  __shared__ double ys[DK][TN + PADT];
  const int64_t bid = blockIdx.x;
  const int64_t row0 = (bid / tiles_n) * TM, col0 = (bid % tiles_n) * TN;
  double val[4][4];
  cov_tile<true, true>(cov, x, n, y, m, d, xx, yy, row0, col0, xs, ys, val);
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = row0 + ty * 4 + i;
    if (r >= n) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t c = col0 + tx * 4 + j;
      if (c < ldo) out[r * ldo + c] = (c < m) ? val[i][j] * w[c] : 0.0;
    }
  }
}

// Cext[j][k] = c_j[dims[k]] (k < nd), Cext[j][nd] = 1, zero up to ldc
__global__ void k_grad_centres(DevCov cov, const double* __restrict__ c, int64_t m, int d, double* __restrict__ cext,
                               int64_t ldc) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= m * ldc) return;
  const int64_t j = idx / ldc;
  const int k = (int)(idx % ldc);
  const DevLeaf lf = cov.leaves[0];
  double v = 0.0;
  if (k < lf.ndims) v = c[j * d + cov.dims[lf.dims_off + k]];
  else if (k == lf.ndims) v = 1.0;
  cext[idx] = v;
}

__global__ void k_grad_combine(DevCov cov, const double* __restrict__ T, int64_t ldt, const double* __restrict__ x,
                               int64_t rows, int d, double* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * d) return;
  out[idx] = 0.0;   // inactive dims
  const int64_t i = idx / d;
  const int dim = (int)(idx % d);
  const DevLeaf lf = cov.leaves[0];
  for (int k = 0; k < lf.ndims; ++k)
    if (cov.dims[lf.dims_off + k] == dim) out[idx] = x[idx] * T[i * ldt + lf.ndims] - T[i * ldt + k];
}

// ---- predictor gradient, composite programs of stationary leaves (Add / Mul / Pow, e.g. the time-sensitive
// product kernel):  grad_i = sum_l sum_j w_j (dP/dk_l)_ij g_l,ij (x_i - c_j)|dims_l.  One covariance-tile pass
// writes Q_l = w_j (dP/dk_l) g_l for every leaf l (the leaf dot products stay in registers, the adjoints come
// from the forward-mode evaluation of the program per element), then one GEMM T_l = Q_l [C|dims_l, 1] per leaf.
__global__ __launch_bounds__(256) void k_grad_coeff_multi(DevCov cov, const double* __restrict__ x, int64_t n,
                                                          const double* __restrict__ y, int64_t m, int d,
                                                          const double* __restrict__ xx, int64_t xx_stride,
                                                          const double* __restrict__ yy,
                                                          const double* __restrict__ w, double* __restrict__ out,
                                                          int64_t ldo, int64_t leaf_stride, int64_t tiles_n) {
  __shared__ double xs[DK][TM + PADT];
  __shared__ double ys[DK][TN + PADT];
  const int64_t bid = blockIdx.x;
  const int64_t row0 = (bid / tiles_n) * TM, col0 = (bid % tiles_n) * TN;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double acc[MLN_MAX_LEAVES][4][4], xr[MLN_MAX_LEAVES][4], yr[MLN_MAX_LEAVES][4];
#pragma unroll
  for (int l = 0; l < MLN_MAX_LEAVES; ++l) {
    if (l >= cov.n_leaves) continue;            // uniform over the workgroup
    leaf_dot(cov, cov.leaves[l], x, n, y, m, d, row0, col0, xs, ys, acc[l]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t r = row0 + ty * 4 + i;
      xr[l][i] = (r < n) ? xx[(int64_t)l * xx_stride + r] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t c = col0 + tx * 4 + j;
      yr[l][j] = (c < m) ? yy[(int64_t)l * m + c] : 0.0;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = row0 + ty * 4 + i;
    if (r >= n) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t c = col0 + tx * 4 + j;
      if (c >= ldo) continue;
      double kv[MLN_MAX_LEAVES], gf[MLN_MAX_LEAVES], a[MLN_MAX_LEAVES];
#pragma unroll
      for (int l = 0; l < MLN_MAX_LEAVES; ++l) {
        kv[l] = 0.0; gf[l] = 0.0;
        if (l >= cov.n_leaves) continue;
        kv[l] = leaf_value(cov.leaves[l], xr[l][i], yr[l][j], acc[l][i][j]);
        gf[l] = leaf_grad_coeff(cov.leaves[l], xr[l][i], yr[l][j], acc[l][i][j]);
      }
      program_adjoints(cov, kv, a);
      const double wc = (c < m) ? w[c] : 0.0;
#pragma unroll
      for (int l = 0; l < MLN_MAX_LEAVES; ++l)
        if (l < cov.n_leaves) out[(int64_t)l * leaf_stride + r * ldo + c] = wc * a[l] * gf[l];
    }
  }
}

// Cext[j][k] = c_j[dims_l[k]] (k < nd_l), Cext[j][nd_l] = 1, zero up to ldc -- for leaf `leaf`
__global__ void k_grad_centres_leaf(DevCov cov, int leaf, const double* __restrict__ c, int64_t m, int d,
                                    double* __restrict__ cext, int64_t ldc) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= m * ldc) return;
  const int64_t j = idx / ldc;
  const int k = (int)(idx % ldc);
  const DevLeaf lf = cov.leaves[leaf];
  double v = 0.0;
  if (k < lf.ndims) v = c[j * d + cov.dims[lf.dims_off + k]];
  else if (k == lf.ndims) v = 1.0;
  cext[idx] = v;
}

// out[i][dims_l[k]] += x_i[dims_l[k]] T_i,nd - T_i,k   (one thread per (i, k); leaves run one after the other)
__global__ void k_grad_combine_leaf(DevCov cov, int leaf, const double* __restrict__ T, int64_t ldt,
                                    const double* __restrict__ x, int64_t rows, int d, double* __restrict__ out) {
  const DevLeaf lf = cov.leaves[leaf];
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * lf.ndims) return;
  const int64_t i = idx / lf.ndims;
  const int k = (int)(idx % lf.ndims);
  const int dim = cov.dims[lf.dims_off + k];
  out[i * d + dim] += x[i * d + dim] * T[i * ldt + lf.ndims] - T[i * ldt + k];
}

// ---- predictor Hessian (base_predictor.py:507-521: jacfwd(jacrev(mean))) --------------------------------------
// With leaf gradients  grad k_l = gam_l (sig_l x - c)|dims_l  (stationary: gam = g, sig = 1; Linear: gam = -1/ls,
// sig = 0) and leaf Hessians  g_l I|dims_l + h_l (x - c)(x - c)^T|dims_l  (stationary only):
//   H_i = sum_j w_j [ sum_l a_l g_l I_l + sum_l (a_l h_l + a_ll gam_l^2) dl dl^T
//                     + sum_{l<l'} a_ll' gam_l gam_l' (dl dl'^T + dl' dl^T) ],   a_l = dP/dk_l, a_ll' = d2P/dk_l dk_l'
// One covariance-tile pass writes the coefficient matrices (Qd_l and one Qp per leaf pair); every pair then is
// ONE GEMM against [vec(c_l c_l'^T) | c_l | c_l' | 1] and an expansion of (sig x - c)(sig' x - c)^T.
__device__ __forceinline__ double leaf_hess_coeff(const DevLeaf& lf, double xx, double yy, double xy) {
  const double inv_ls = lf.alpha_inv_ls[1];
  const double sq = xx - 2.0 * xy + yy + 1e-12;
  if (!(sq > 0.0) || lf.kind == MLN_K_LINEAR) return 0.0;
  const double dist = sqrt(sq), il2 = inv_ls * inv_ls;
  switch (lf.kind) {
    case MLN_K_MATERN32: {                  // g = -f^2 e^{-f r}
      const double f = 1.7320508075688772 * inv_ls;
      return f * f * f * exp(-f * dist) / dist;
    }
    case MLN_K_MATERN52: {                  // g = -(f^2 / 3)(1 + f r) e^{-f r}
      const double f = 2.23606797749979 * inv_ls, f2 = f * f;
      return f2 * f2 * exp(-f * dist) * 0.3333333333333333;
    }
    case MLN_K_EXPQUAD: {                   // g = -k / ls^2
      const double r = dist * inv_ls;
      return exp(-0.5 * (r * r)) * il2 * il2;
    }
    case MLN_K_EXPONENTIAL: {               // g = -k / (2 ls r)
      const double e = exp(-0.5 * dist * inv_ls);
      return e * (0.25 * il2 / sq + 0.5 * inv_ls / (sq * dist));
    }
    default: {                              // RatQuad: g = -b^{-alpha-1} / ls^2
      const double r = dist * inv_ls, b = r * r / (2.0 * lf.alpha) + 1.0;
      return (lf.alpha + 1.0) / lf.alpha * pow(b, -lf.alpha - 2.0) * il2 * il2;
    }
  }
}

__host__ __device__ inline int hess_pair_slot(int L, int l, int lp) {   // l <= lp: slots after the L diagonal ones
  return L + l * L - (l * (l - 1)) / 2 + (lp - l);
}

__global__ __launch_bounds__(256) void k_hess_coeff(DevCov cov, const double* __restrict__ x, int64_t n,
                                                    const double* __restrict__ y, int64_t m, int d,
                                                    const double* __restrict__ xx, int64_t xx_stride,
                                                    const double* __restrict__ yy,
                                                    const double* __restrict__ w, double* __restrict__ out,
                                                    int64_t ldo, int64_t slot_stride, int64_t tiles_n) {
  __shared__ double xs[DK][TM + PADT];
  __shared__ double ys[DK][TN + PADT];
  const int64_t bid = blockIdx.x;
  const int64_t row0 = (bid / tiles_n) * TM, col0 = (bid % tiles_n) * TN;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int L = cov.n_leaves;
  double acc[MLN_MAX_LEAVES][4][4], xr[MLN_MAX_LEAVES][4], yr[MLN_MAX_LEAVES][4];
#pragma unroll
  for (int l = 0; l < MLN_MAX_LEAVES; ++l) {
    if (l >= L) continue;
    leaf_dot(cov, cov.leaves[l], x, n, y, m, d, row0, col0, xs, ys, acc[l]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t r = row0 + ty * 4 + i;
      xr[l][i] = (r < n) ? xx[(int64_t)l * xx_stride + r] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t c = col0 + tx * 4 + j;
      yr[l][j] = (c < m) ? yy[(int64_t)l * m + c] : 0.0;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = row0 + ty * 4 + i;
    if (r >= n) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t c = col0 + tx * 4 + j;
      if (c >= ldo) continue;
      double kv[MLN_MAX_LEAVES], gam[MLN_MAX_LEAVES], gd[MLN_MAX_LEAVES], hc[MLN_MAX_LEAVES];
      double a1[MLN_MAX_LEAVES], a2[MLN_MAX_LEAVES][MLN_MAX_LEAVES];
#pragma unroll
      for (int l = 0; l < MLN_MAX_LEAVES; ++l) {
        kv[l] = 0.0; gam[l] = 0.0; gd[l] = 0.0; hc[l] = 0.0;
        if (l >= L) continue;
        const DevLeaf lf = cov.leaves[l];
        kv[l] = leaf_value(lf, xr[l][i], yr[l][j], acc[l][i][j]);
        if (lf.kind == MLN_K_LINEAR) { gam[l] = -lf.alpha_inv_ls[1]; continue; }
        gam[l] = gd[l] = leaf_grad_coeff(lf, xr[l][i], yr[l][j], acc[l][i][j]);
        hc[l] = leaf_hess_coeff(lf, xr[l][i], yr[l][j], acc[l][i][j]);
      }
      program_second(cov, kv, a1, a2);
      const double wc = (c < m) ? w[c] : 0.0;
      const int64_t at = r * ldo + c;
#pragma unroll
      for (int l = 0; l < MLN_MAX_LEAVES; ++l) {
        if (l >= L) continue;
        out[(int64_t)l * slot_stride + at] = wc * a1[l] * gd[l];
#pragma unroll
        for (int lp = l; lp < MLN_MAX_LEAVES; ++lp) {
          if (lp >= L) continue;
          const double v = (lp == l) ? a1[l] * hc[l] + a2[l][l] * gam[l] * gam[l] : a2[l][lp] * gam[l] * gam[lp];
          out[(int64_t)hess_pair_slot(L, l, lp) * slot_stride + at] = wc * v;
        }
      }
    }
  }
}

// Cp[j] = [ c_j[dl[a]] c_j[dlp[b]] (a * nlp + b) | c_j[dl[a]] | c_j[dlp[b]] | 1 ], zero up to ldc
__global__ void k_hess_centres(DevCov cov, int l, int lp, const double* __restrict__ c, int64_t m, int d,
                               double* __restrict__ cp, int64_t ldc) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= m * ldc) return;
  const int64_t j = idx / ldc;
  const int k = (int)(idx % ldc);
  const DevLeaf la = cov.leaves[l], lb = cov.leaves[lp];
  const int na = la.ndims, nb = lb.ndims;
  const double* cj = c + j * d;
  double v = 0.0;
  if (k < na * nb) v = cj[cov.dims[la.dims_off + k / nb]] * cj[cov.dims[lb.dims_off + k % nb]];
  else if (k < na * nb + na) v = cj[cov.dims[la.dims_off + (k - na * nb)]];
  else if (k < na * nb + na + nb) v = cj[cov.dims[lb.dims_off + (k - na * nb - na)]];
  else if (k == na * nb + na + nb) v = 1.0;
  cp[idx] = v;
}

// H_i[dl[a]][dlp[b]] (transpose = 0) or H_i[dlp[b]][dl[a]] (transpose = 1)
//   += sig sig' s x_a x_b - sig x_a (Q c')_b - sig' (Q c)_a x_b + (Q c c'^T)_ab
__global__ void k_hess_combine(DevCov cov, int l, int lp, int transpose, const double* __restrict__ T, int64_t ldt,
                               const double* __restrict__ x, int64_t rows, int d, double* __restrict__ H) {
  const DevLeaf la = cov.leaves[l], lb = cov.leaves[lp];
  const int na = la.ndims, nb = lb.ndims;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * na * nb) return;
  const int64_t i = idx / (na * nb);
  const int ab = (int)(idx % (na * nb)), a = ab / nb, b = ab % nb;
  const int da = cov.dims[la.dims_off + a], db = cov.dims[lb.dims_off + b];
  const double sa = (la.kind == MLN_K_LINEAR) ? 0.0 : 1.0, sb = (lb.kind == MLN_K_LINEAR) ? 0.0 : 1.0;
  const double* Ti = T + i * ldt;
  const double xa = x[i * d + da], xb = x[i * d + db];
  const double val = sa * sb * Ti[na * nb + na + nb] * xa * xb - sa * xa * Ti[na * nb + na + b]
                     - sb * Ti[na * nb + a] * xb + Ti[ab];
  double* Hi = H + i * (int64_t)d * d;
  if (transpose) Hi[db * d + da] += val; else Hi[da * d + db] += val;
}

// H_i[dim][dim] += sum_j Qd[i][j] for the dims of one stationary leaf
__global__ __launch_bounds__(256) void k_hess_diag(DevCov cov, int l, const double* __restrict__ Qd, int64_t ldq,
                                                   int64_t m, int64_t rows, int d, double* __restrict__ H) {
  __shared__ double red[256];
  const int64_t i = blockIdx.x;
  if (i >= rows) return;
  double s = 0.0;
  for (int64_t j = threadIdx.x; j < m; j += 256) s += Qd[i * ldq + j];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  const DevLeaf lf = cov.leaves[l];
  if (lf.kind == MLN_K_LINEAR) return;
  for (int k = threadIdx.x; k < lf.ndims; k += 256) {
    const int dim = cov.dims[lf.dims_off + k];
    H[i * (int64_t)d * d + (int64_t)dim * d + dim] += red[0];
  }
}

// mean_i = mu + sum_j cov(x_i, y_j) w_j   (conditional.py:899-906); K never leaves registers.
template <bool SINGLE>
__global__ __launch_bounds__(256) void k_predict_mean1(DevCov cov, const double* __restrict__ x, int64_t n,
                                                       const double* __restrict__ y, int64_t m, int d,
                                                       const double* __restrict__ xx,
                                                       const double* __restrict__ yy,
                                                       const double* __restrict__ w, double mu,
                                                       double* __restrict__ out) {
  __shared__ double xs[DK][TM + PADT];
  __shared__ double ys[DK][TN + PADT];
  __shared__ double red[TM][17];
  const int64_t row0 = (int64_t)blockIdx.x * TM;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double part[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t col0 = 0; col0 < m; col0 += TN) {
    double val[4][4];
    cov_tile<SINGLE>(cov, x, n, y, m, d, xx, yy, row0, col0, xs, ys, val);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int64_t c = col0 + tx * 4 + j;
      double wj = (c < m) ? w[c] : 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) part[i] = fma(val[i][j], wj, part[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) red[ty * 4 + i][tx] = part[i];
  __syncthreads();
  if (threadIdx.x < TM) {
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += red[threadIdx.x][t];
    int64_t r = row0 + threadIdx.x;
    if (r < n) out[r] = mu + s;
  }
}

// Fused predictive mean on the matrix cores (single leaf over all d columns, one output): the tile scheme of
// k_kernel_matrix_mfma, but a wave keeps its 16 rows and walks ALL centre tiles, multiplying each covariance
// value by its weight and summing along the row -- the n' x m matrix never exists (conditional.py:899-906).
__global__ __launch_bounds__(256) void k_predict_mean_mfma(DevCov cov, const double* __restrict__ x, int64_t n,
                                                           const double* __restrict__ y, int64_t m, int d,
                                                           const double* __restrict__ xx,
                                                           const double* __restrict__ yy,
                                                           const double* __restrict__ w, double mu,
                                                           double* __restrict__ out) {
  const DevLeaf lf = cov.leaves[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * TM + wave * 16;
  const int64_t ar = (row0 + li < n) ? row0 + li : n - 1;
  const double* __restrict__ xa = x + ar * d;
  double xr[4], part[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = row0 + lk + 4 * r;
    xr[r] = (row < n) ? xx[row] : 0.0;
    part[r] = 0.0;
  }
  for (int64_t col0 = 0; col0 < m; col0 += TN) {
    const double* yb[4];
    double yc[4], wc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int64_t c = col0 + 16 * t + li;
      const int64_t cc = (c < m) ? c : m - 1;
      yb[t] = y + cc * d;
      yc[t] = yy[cc];
      wc[t] = (c < m) ? w[cc] : 0.0;
    }
    v4d_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = v4d_t{0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < d; k0 += 4) {
      const int k = k0 + lk;
      const int kc = (k < d) ? k : d - 1;
      const double mask = (k < d) ? 1.0 : 0.0;
      const double a = xa[kc] * mask;
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, yb[t][kc], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[r] = fma(leaf_value(lf, xr[r], yc[t], acc[t][r]), wc[t], part[r]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double s_ = part[r];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s_ += __shfl_xor(s_, off, 64);   // over the 16 columns (li) of a row
    const int64_t row = row0 + lk + 4 * r;
    if (li == 0 && row < n) out[row] = mu + s_;
  }
}

// xx[leaf][i] = sum over the leaf's active dims of x_i^2   (util.py:362).  Eight lanes per row: a wave's load covers eight
// consecutive rows (3.2 KB of consecutive lines at d = 50) instead of 64 addresses d doubles apart -- one lane per row
// read the 0.4 GB of C3's cells at 0.47 TB/s (0.85 ms per fit).
__global__ __launch_bounds__(256) void k_row_sqnorms(DevCov cov, const double* __restrict__ x, int64_t n, int d,
                                                     double* __restrict__ xx) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const int q = threadIdx.x & 7;
  const bool ok = i < n;
  const double* row = x + (ok ? i : 0) * (int64_t)d;
  for (int l = 0; l < cov.n_leaves; ++l) {
    const DevLeaf lf = cov.leaves[l];
    double s = 0.0;
    for (int k = q; k < lf.ndims; k += 8) {
      const double v = row[cov.dims[lf.dims_off + k]];
      s = fma(v, v, s);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (ok && q == 0) xx[(int64_t)l * n + i] = s;
  }
}


// The search compares |x|^2 - 2 x.y + |y|^2, whose absolute error is ~eps |x|^2: enough to pick the winner (up to ties
// inside that noise), not to report a tiny distance -- a duplicated cell would come out as ~1e-7 |x| instead of 0, and the
// reference replaces exactly the non-positive distances (validation.py:528-592).  The reported value is therefore
// recomputed from the winner's coordinates: sum (x_k - y_k)^2, relative error ~eps.
__device__ __forceinline__ double nn_direct_distance(const double* __restrict__ x, int64_t i, const double* __restrict__ y,
                                                     int64_t j, int d) {
  if (j < 0) return INFINITY;
  double s = 0.0;
  for (int k = 0; k < d; ++k) {
    const double t = x[i * d + k] - y[j * d + k];
    s = fma(t, t, s);
  }
  return sqrt(s);
}

// Exact Euclidean nearest-neighbour distance of every row of x among the rows of y, skipping the
// pair (i, i + self_offset).  Replaces the approximate pynndescent search of the reference
// (mellon/parameters.py:352-433) -- same tile structure as the covariance kernel; the running
// minimum of the squared distance stays in registers across all centre tiles.
__global__ __launch_bounds__(256) void k_nn_distances(const double* __restrict__ x, int64_t n,
                                                      const double* __restrict__ y, int64_t m, int d,
                                                      const double* __restrict__ xx,
                                                      const double* __restrict__ yy, int64_t self_offset,
                                                      const int64_t* __restrict__ excl, double* __restrict__ out) {
  __shared__ double xs[DK][TM + PADT];
  __shared__ double ys[DK][TN + PADT];
  __shared__ double red[TM][17];
  __shared__ int64_t redj[TM][17];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * TM;
  double best[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
  int64_t bj[4] = {-1, -1, -1, -1};
  double xr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int64_t r = row0 + ty * 4 + i;
    xr[i] = (r < n) ? xx[r] : 0.0;
  }
  for (int64_t col0 = 0; col0 < m; col0 += TN) {
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    for (int k0 = 0; k0 < d; k0 += DK) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int idx = tid + 256 * q;
        int r = idx >> 4, k = idx & 15;
        double vx = 0.0, vy = 0.0;
        if (k0 + k < d) {
          if (row0 + r < n) vx = x[(row0 + r) * (int64_t)d + k0 + k];
          if (col0 + r < m) vy = y[(col0 + r) * (int64_t)d + k0 + k];
        }
        xs[k][r] = vx;
        ys[k][r] = vy;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < DK; ++k) {
        double a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = xs[k][ty * 4 + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = ys[k][tx * 4 + j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t c = col0 + tx * 4 + j;
      const double yj = (c < m) ? yy[c] : 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t r = row0 + ty * 4 + i;
        const double sq = xr[i] - 2.0 * acc[i][j] + yj;
        const int64_t skip = (excl && r < n) ? excl[r] : r + self_offset;
        if (c < m && c != skip && sq < best[i]) { best[i] = sq; bj[i] = c; }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { red[ty * 4 + i][tx] = best[i]; redj[ty * 4 + i][tx] = bj[i]; }
  __syncthreads();
  if (tid < TM) {
    double s = INFINITY;
    int64_t j = -1;
#pragma unroll
    for (int t = 0; t < 16; ++t)
      if (red[tid][t] < s) { s = red[tid][t]; j = redj[tid][t]; }
    int64_t r = row0 + tid;
    if (r < n) out[r] = nn_direct_distance(x, r, y, j, d);
  }
}

// Exact 1-NN on the matrix cores (d <= 64): a workgroup of 8 waves owns 128 query rows, every wave keeps the
// MFMA A operands of its 16 rows in registers for the whole pass and walks all tiles of 64 candidate rows,
// which are staged through LDS once per workgroup (double-buffered, coalesced in, [row][k] with a
// 68-double row stride out) and shared by the 8 waves.  Only xx - 2 x.y + yy and a running minimum per
// element: the pass is bound by the fp64 matrix pipe (2 n m d flops).

__global__ __launch_bounds__(512) void k_nn_distances_mfma(const double* __restrict__ x, int64_t n,
                                                           const double* __restrict__ y, int64_t m, int d,
                                                           const double* __restrict__ xx,
                                                           const double* __restrict__ yy, int64_t self_offset,
                                                           const int64_t* __restrict__ excl, double* __restrict__ out) {
  __shared__ double ys[2][TN * NNS];
  __shared__ double yn[2][TN];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 16;
  const int ksteps = (d + 3) / 4;
  // A operands of this wave's 16 rows: a[ks] = x[row0 + li][4 ks + lk]
  double a[16];
  {
    const int64_t ar = (row0 + li < n) ? row0 + li : n - 1;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int k = 4 * ks + lk;
      a[ks] = (ks < ksteps && k < d) ? x[ar * d + k] : 0.0;
    }
  }
  double xr[4], best[4];
  int bt[4];            // the winner so far as a 16-column group: column = 16 bt + li (bt < 2^31: m < 2^35)
  int64_t ex[4];        // the candidate that does not count for each of this lane's rows
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = row0 + lk + 4 * r;
    xr[r] = (row < n) ? xx[row] : 0.0;
    ex[r] = (excl && row < n) ? excl[row] : row + self_offset;
    best[r] = INFINITY;
    bt[r] = -1;
  }
  for (int e = tid; e < 2 * TN * NNS; e += 512) (&ys[0][0])[e] = 0.0;   // zero incl. the k padding
  __syncthreads();
  auto stage = [&](int buf, int64_t col0) {
    const int cnt = TN * d;
    for (int e = tid; e < cnt; e += 512) {
      const int r = e / d, k = e - r * d;
      ys[buf][r * NNS + k] = (col0 + r < m) ? y[(col0 + r) * d + k] : 0.0;
    }
    if (tid < TN) yn[buf][tid] = (col0 + tid < m) ? yy[col0 + tid] : 0.0;
  };
  stage(0, 0);
  __syncthreads();
  int buf = 0;
  for (int64_t col0 = 0; col0 < m; col0 += TN, buf ^= 1) {
    if (col0 + TN < m) stage(buf ^ 1, col0 + TN);
    v4d_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = v4d_t{0.0, 0.0, 0.0, 0.0};
    const double* yb = &ys[buf][li * NNS + lk];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks < ksteps) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], yb[16 * t * NNS + 4 * ks], acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int64_t c = col0 + 16 * t + li;
      const double yc = yn[buf][16 * t + li];
      const int grp = (int)(col0 >> 4) + t;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double sq = xr[r] - 2.0 * acc[t][r] + yc;
        if (c < m && c != ex[r] && sq < best[r]) { best[r] = sq; bt[r] = grp; }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double s_ = best[r];
    int64_t j_ = bt[r] < 0 ? -1 : 16 * (int64_t)bt[r] + li;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      const double so = __shfl_xor(s_, off, 64);
      const int64_t jo = __shfl_xor(j_, off, 64);
      if (so < s_ || (so == s_ && jo >= 0 && (j_ < 0 || jo < j_))) { s_ = so; j_ = jo; }
    }
    const int64_t row = row0 + lk + 4 * r;
    if (li == 0 && row < n) out[row] = nn_direct_distance(x, row, y, j_, d);
  }
}

__global__ void k_row_sqnorms_all(const double* __restrict__ x, int64_t n, int d, double* __restrict__ xx) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int k = 0; k < d; ++k) {
    double v = x[i * (int64_t)d + k];
    s = fma(v, v, s);
  }
  xx[i] = s;
}

// diag_i = cov(x_i, x_i) (reference base_cov.py:71-93): the whole program with xy = xx = yy per leaf.
__global__ void k_cov_diag(DevCov cov, const double* __restrict__ x, int64_t n, int d, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  int sp = 0;
  for (int t = 0; t < cov.n_toks; ++t) {
    const int op = cov.tok_op[t];
    if (op == MLN_OP_LEAF || op == MLN_OP_CONST) {
      double v;
      if (op == MLN_OP_CONST) {
        v = cov.tok_val[t];
      } else {
        const DevLeaf lf = cov.leaves[cov.tok_leaf[t]];
        double xx = 0.0;
        for (int k = 0; k < lf.ndims; ++k) {
          const double q = x[i * (int64_t)d + cov.dims[lf.dims_off + k]];
          xx = fma(q, q, xx);
        }
        v = leaf_value(lf, xx, xx, xx);
      }
      if (sp > 0) { s2 = s1; s1 = s0; }
      s0 = v;
      ++sp;
    } else {
      const double l = s1, r = s0;
      s0 = (op == MLN_OP_ADD) ? (l + r) : (op == MLN_OP_MUL) ? (l * r) : pow(l, r);
      s1 = s2;
      --sp;
    }
  }
  out[i] = s0;
}

// out_i = base_i + sign * sum_j T_ij^2 [* scale_j^2]
__global__ void k_row_sumsq(const double* __restrict__ T, int64_t ld, int64_t rows, int64_t cols,
                            const double* __restrict__ base, double sign, double* __restrict__ out) {
  const int64_t i = blockIdx.x;
  if (i >= rows) return;
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t j = threadIdx.x; j < cols; j += 256) { const double v = T[i * ld + j]; s = fma(v, v, s); }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[i] = (base ? base[i] : 0.0) + sign * red[0];
}

int sqnorms(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, int d, double* xx) {
  if (n == 0) return MLN_OK;
  hipLaunchKernelGGL(k_row_sqnorms, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, ctx->stream, cov, x, n, d, xx);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

}  // namespace

// The persistent-row kernels (cov_rows_impl.h, kernel_rows_prod_impl.h, predict_rows*.hip) stage centre tiles without
// bounds checks: they are handed a COPY of the centres followed by ROWS_PAD zero rows (and norms / weights padded alike).
constexpr int64_t ROWS_PAD = 3 * 64;   // (a predict kernel requests tile t + 2 up to t = ceil(m / 64) - 1)
static int pad_rows(mln_ctx* ctx, const double* src, int64_t rows, int64_t cols, double* dst) {
  MLN_HIP(ctx, hipMemcpyAsync(dst, src, sizeof(double) * (size_t)(rows * cols), hipMemcpyDeviceToDevice, ctx->stream));
  MLN_HIP(ctx, hipMemsetAsync(dst + rows * cols, 0, sizeof(double) * (size_t)(ROWS_PAD * cols), ctx->stream));
  return MLN_OK;
}

int launch_kernel_matrix(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y,
                         int64_t m, int d, double* out, int64_t ldo, double add_diag, float* out32, int q32) {
  if (n == 0 || m == 0) return MLN_OK;
  double* norms = nullptr;
  const int64_t mp = m + ROWS_PAD;
  MLN_TRY(mln_scratch(ctx, sizeof(double) * ((size_t)cov.n_leaves * (size_t)(n + mp) + (size_t)(mp * d)), (void**)&norms));
  double* xx = norms;
  double* yy = norms + (int64_t)cov.n_leaves * n;
  double* ypad = yy + (int64_t)cov.n_leaves * mp;
  MLN_TRY(sqnorms(ctx, cov, x, n, d, xx));
  MLN_TRY(sqnorms(ctx, cov, y, m, d, yy));
  const int64_t tiles_n = (ldo + TN - 1) / TN, tiles_m = (n + TM - 1) / TM;   // covers the pad columns too
  const int64_t nblk = tiles_n * tiles_m;
  if (nblk > 0x7fffffffLL) { mln_set_error(ctx, "kernel matrix too large for one launch"); return MLN_ERR_UNSUPPORTED; }
  const bool single = (cov.n_toks == 1);
  bool contiguous = single && cov.leaves[0].ndims == d;
  if (contiguous)
    for (int k = 0; k < d; ++k) contiguous = contiguous && cov.dims[cov.leaves[0].dims_off + k] == k;
  static const bool no_rows_env = mln_experiment("MELLON_AMD_KM_NO_ROWS") != nullptr;
  // (the persistent-row kernels read the centres WITHOUT bounds checks up to tile ceil(ldo / 64) + 1: the zero rows pad_rows
  //  appends cover that only while the output's leading dimension stays within 64 columns of m -- every caller passes
  //  ldo = m or pad16(m); anything wider takes the tiled kernels)
  const bool no_rows = no_rows_env || ldo > m + 64;
  if (contiguous && !no_rows && d <= 64 && n >= 4096 && m >= 256 && cov.leaves[0].kind != MLN_K_LINEAR &&
      cov.leaves[0].kind != MLN_K_DISTANCE && cov.leaves[0].kind != MLN_K_RATQUAD && (!out32 || q32)) {
    MLN_TRY(pad_rows(ctx, y, m, d, ypad));
    MLN_HIP(ctx, hipMemsetAsync(yy + m, 0, sizeof(double) * (size_t)ROWS_PAD, ctx->stream));   // norms of the pad rows
    MLN_TRY(launch_kernel_matrix_rows_q(ctx, cov, x, n, ypad, m, d, xx, yy, out, ldo, add_diag, out32, q32));
  } else if (!no_rows && n >= 4096 && m >= 256 && (!out32 || q32) && predict_rows_prod_eligible(cov, d)) {
    // the time-sensitive product kernel: state leaf x time leaf, both in the persistent-row kernel (leaf 0's norms come
    // first; leaf 1's, which it does not use, are overwritten by the zero padding)
    MLN_TRY(pad_rows(ctx, y, m, d, ypad));
    MLN_HIP(ctx, hipMemsetAsync(yy + m, 0, sizeof(double) * (size_t)ROWS_PAD, ctx->stream));
    MLN_TRY(launch_kernel_matrix_rows_prod(ctx, cov, x, n, ypad, m, d, xx, yy, out, ldo, add_diag, out32, q32));
  } else if (contiguous && n * m >= 4096)
    hipLaunchKernelGGL(k_kernel_matrix_mfma, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, cov, x, n, y, m, d,
                       xx, yy, out, ldo, add_diag, tiles_n, out32, q32);
  else if (single)
    hipLaunchKernelGGL(k_kernel_matrix<true>, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, cov, x, n, y, m, d,
                       xx, yy, out, ldo, add_diag, tiles_n, out32, q32);
  else
    hipLaunchKernelGGL(k_kernel_matrix<false>, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, cov, x, n, y, m, d,
                       xx, yy, out, ldo, add_diag, tiles_n, out32, q32);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_predict_mean1(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y,
                         int64_t m, int d, const double* w, double mu, double* out) {
  if (n == 0) return MLN_OK;
  double* norms = nullptr;
  const int64_t mp = m + ROWS_PAD;
  MLN_TRY(mln_scratch(ctx, sizeof(double) * ((size_t)cov.n_leaves * (size_t)(n + mp) + (size_t)(mp * (d + 1))), (void**)&norms));
  double* xx = norms;
  double* yy = norms + (int64_t)cov.n_leaves * n;
  double* ypad = yy + (int64_t)cov.n_leaves * mp;
  double* wpad = ypad + mp * d;
  MLN_TRY(sqnorms(ctx, cov, x, n, d, xx));
  MLN_TRY(sqnorms(ctx, cov, y, m, d, yy));
  const int64_t nblk = (n + TM - 1) / TM;
  const bool single = (cov.n_toks == 1);
  bool contiguous = single && cov.leaves[0].ndims == d;
  if (contiguous)
    for (int k = 0; k < d; ++k) contiguous = contiguous && cov.dims[cov.leaves[0].dims_off + k] == k;
  if (contiguous && d <= 64 && n >= 4096 && m >= 256 && cov.leaves[0].kind != MLN_K_LINEAR &&
      cov.leaves[0].kind != MLN_K_DISTANCE) {
    MLN_TRY(pad_rows(ctx, y, m, d, ypad));
    MLN_TRY(pad_rows(ctx, w, m, 1, wpad));
    MLN_HIP(ctx, hipMemsetAsync(yy + m, 0, sizeof(double) * (size_t)ROWS_PAD, ctx->stream));
    MLN_TRY(launch_predict_mean_rows(ctx, cov, x, n, ypad, m, d, xx, yy, wpad, mu, out));
  } else if (n >= 4096 && m >= 256 && predict_rows_prod_eligible(cov, d)) {
    MLN_TRY(pad_rows(ctx, y, m, d, ypad));
    MLN_TRY(pad_rows(ctx, w, m, 1, wpad));
    MLN_HIP(ctx, hipMemsetAsync(yy + m, 0, sizeof(double) * (size_t)ROWS_PAD, ctx->stream));   // (leaf 0's norms come first)
    MLN_TRY(launch_predict_mean_rows_prod(ctx, cov, x, n, ypad, m, d, xx, yy, wpad, mu, out));
  } else if (contiguous && n * m >= 4096)
    hipLaunchKernelGGL(k_predict_mean_mfma, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, cov, x, n, y, m, d,
                       xx, yy, w, mu, out);
  else if (single)
    hipLaunchKernelGGL(k_predict_mean1<true>, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, cov, x, n, y, m, d,
                       xx, yy, w, mu, out);
  else
    hipLaunchKernelGGL(k_predict_mean1<false>, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, cov, x, n, y, m, d,
                       xx, yy, w, mu, out);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_nn_distances_exact(mln_ctx* ctx, const double* x, int64_t n, const double* y, int64_t m, int d,
                              int64_t self_offset, const int64_t* excl, double* out) {
  if (n == 0) return MLN_OK;
  double* norms = nullptr;
  MLN_HIP(ctx, mln_dmalloc((void**)&norms, sizeof(double) * (size_t)(n + m)));
  double* xx = norms;
  double* yy = norms + n;
  hipLaunchKernelGGL(k_row_sqnorms_all, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, x, n, d, xx);
  hipLaunchKernelGGL(k_row_sqnorms_all, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, y, m, d, yy);
  if (d <= 64 && n * m >= 4096)
    hipLaunchKernelGGL(k_nn_distances_mfma, dim3((unsigned)((n + 127) / 128)), dim3(512), 0, ctx->stream, x, n, y, m, d,
                       xx, yy, self_offset, excl, out);
  else
    hipLaunchKernelGGL(k_nn_distances, dim3((unsigned)((n + TM - 1) / TM)), dim3(256), 0, ctx->stream, x, n, y, m, d,
                       xx, yy, self_offset, excl, out);
  hipError_t e = hipGetLastError();
  (void)hipStreamSynchronize(ctx->stream);
  (void)mln_dfree(norms);
  if (e != hipSuccess) return mln_hip_fail(ctx, e, "nn_distances", __FILE__, __LINE__);
  return MLN_OK;
}

// Exact nearest-neighbour distances.  Large searches (d <= 64) go through the fp16 pre-filter with fp64 certification
// (rowmin_f16.hip): same result, a fraction of the time.  MELLON_AMD_NN_PREFILTER=0 forces the plain fp64 search.
int launch_nn_distances(mln_ctx* ctx, const double* x, int64_t n, const double* y, int64_t m, int d,
                        int64_t self_offset, double* out) {
  if (n == 0) return MLN_OK;
  const char* e_on = std::getenv("MELLON_AMD_NN_PREFILTER");          // read per call: the tests flip it
  const char* e_min = std::getenv("MELLON_AMD_NN_PREFILTER_MIN");
  const bool prefilter = !(e_on && std::atoi(e_on) == 0);
  const int64_t min_pairs = e_min ? std::atoll(e_min) : ((int64_t)1 << 26);
  if (prefilter && d <= 64 && m >= 2 && n * m >= min_pairs && m < 2147483647LL)
    return nn_distances_prefiltered(ctx, x, n, y, m, d, self_offset, out, nullptr);
  return launch_nn_distances_exact(ctx, x, n, y, m, d, self_offset, nullptr, out);
}

int launch_cov_diag(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, int d, double* out) {
  if (n == 0) return MLN_OK;
  hipLaunchKernelGGL(k_cov_diag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, cov, x, n, d, out);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_row_sumsq(mln_ctx* ctx, const double* T, int64_t ld, int64_t rows, int64_t cols, const double* base,
                     double sign, double* out) {
  if (rows == 0) return MLN_OK;
  hipLaunchKernelGGL(k_row_sumsq, dim3((unsigned)rows), dim3(256), 0, ctx->stream, T, ld, rows, cols, base, sign, out);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}


// Hessian of the predicted mean at every row of x (see k_hess_coeff): out is n x d x d.
int launch_predict_hessian(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* c, int64_t m,
                           int d, const double* w, double* out) {
  if (n == 0) return MLN_OK;
  const int L = cov.n_leaves;
  const int n_slots = L + (L * (L + 1)) / 2;
  const int64_t ldq = ((m + 15) / 16) * 16;
  int64_t chunk = ((int64_t)1 << 28) / (ldq * n_slots);        // <= 2 GiB of coefficient matrices per chunk
  if (chunk > 16384) chunk = 16384;
  if (chunk < 64) chunk = 64;
  if (chunk > n) chunk = n;
  // centre operands of the pair GEMMs
  std::vector<int64_t> cp_off((size_t)L * L, 0), cp_ld((size_t)L * L, 0);
  int64_t cp_total = 0, ldt_max = 0;
  for (int l = 0; l < L; ++l)
    for (int lp = l; lp < L; ++lp) {
      const int64_t ncol = (int64_t)cov.leaves[l].ndims * cov.leaves[lp].ndims + cov.leaves[l].ndims + cov.leaves[lp].ndims + 1;
      const int64_t ld = ((ncol + 15) / 16) * 16;
      cp_off[(size_t)l * L + lp] = cp_total; cp_ld[(size_t)l * L + lp] = ld;
      cp_total += m * ld;
      if (ld > ldt_max) ldt_max = ld;
    }
  double *norms = nullptr, *Q = nullptr, *T = nullptr, *cp = nullptr;
  MLN_TRY(mln_scratch(ctx, sizeof(double) * (size_t)L * (size_t)(n + m), (void**)&norms));
  double* xx = norms;
  double* yy = norms + (size_t)L * n;
  MLN_TRY(sqnorms(ctx, cov, x, n, d, xx));
  MLN_TRY(sqnorms(ctx, cov, c, m, d, yy));
  const int64_t slot_stride = chunk * ldq;
  MLN_HIP(ctx, mln_dmalloc((void**)&Q, sizeof(double) * (size_t)n_slots * slot_stride));
  MLN_HIP(ctx, mln_dmalloc((void**)&T, sizeof(double) * (size_t)chunk * ldt_max));
  MLN_HIP(ctx, mln_dmalloc((void**)&cp, sizeof(double) * (size_t)cp_total));
  for (int l = 0; l < L; ++l)
    for (int lp = l; lp < L; ++lp) {
      const int64_t ld = cp_ld[(size_t)l * L + lp];
      hipLaunchKernelGGL(k_hess_centres, dim3((unsigned)((m * ld + 255) / 256)), dim3(256), 0, ctx->stream, cov, l, lp, c, m,
                         d, cp + cp_off[(size_t)l * L + lp], ld);
    }
  int rc = MLN_OK;
  hipError_t e0 = hipMemsetAsync(out, 0, sizeof(double) * (size_t)n * d * d, ctx->stream);
  if (e0 != hipSuccess) rc = mln_hip_fail(ctx, e0, "predict_hessian", __FILE__, __LINE__);
  const int64_t tiles_n = (ldq + TN - 1) / TN;
  for (int64_t r0 = 0; r0 < n && rc == MLN_OK; r0 += chunk) {
    const int64_t rows = (n - r0 < chunk) ? (n - r0) : chunk;
    const int64_t nblk = tiles_n * ((rows + TM - 1) / TM);
    double* Hc = out + r0 * (int64_t)d * d;
    hipLaunchKernelGGL(k_hess_coeff, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, cov, x + r0 * d, rows, c, m, d,
                       xx + r0, n, yy, w, Q, ldq, slot_stride, tiles_n);
    for (int l = 0; l < L && rc == MLN_OK; ++l) {
      hipLaunchKernelGGL(k_hess_diag, dim3((unsigned)rows), dim3(256), 0, ctx->stream, cov, l, Q + (size_t)l * slot_stride,
                         ldq, m, rows, d, Hc);
      for (int lp = l; lp < L && rc == MLN_OK; ++lp) {
        const int na = cov.leaves[l].ndims, nb = cov.leaves[lp].ndims;
        const int64_t ld = cp_ld[(size_t)l * L + lp];
        GemmArgs g{};
        g.A = Q + (size_t)hess_pair_slot(L, l, lp) * slot_stride; g.lda = ldq; g.ta = 0;
        g.B = cp + cp_off[(size_t)l * L + lp]; g.ldb = ld; g.tb = 0; g.C = T; g.ldc = ld;
        g.M = rows; g.N = (int64_t)na * nb + na + nb + 1; g.K = m; g.alpha = 1.0; g.beta = 0.0; g.split_k = 1;
        rc = launch_dgemm(ctx, g);
        if (rc != MLN_OK) break;
        const unsigned nb_ = (unsigned)((rows * na * nb + 255) / 256);
        hipLaunchKernelGGL(k_hess_combine, dim3(nb_), dim3(256), 0, ctx->stream, cov, l, lp, 0, T, ld, x + r0 * d, rows, d, Hc);
        if (lp != l)
          hipLaunchKernelGGL(k_hess_combine, dim3(nb_), dim3(256), 0, ctx->stream, cov, l, lp, 1, T, ld, x + r0 * d, rows, d, Hc);
      }
    }
  }
  hipError_t e = hipGetLastError();
  (void)hipStreamSynchronize(ctx->stream);
  (void)mln_dfree(Q); (void)mln_dfree(T); (void)mln_dfree(cp);
  if (rc == MLN_OK && e != hipSuccess) rc = mln_hip_fail(ctx, e, "predict_hessian", __FILE__, __LINE__);
  return rc;
}

// Predictor gradient of a composite program of stationary leaves through the matrix cores (see k_grad_coeff_multi).
int launch_predict_gradient_gemm_multi(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* c,
                                       int64_t m, int d, const double* w, double* out) {
  if (n == 0) return MLN_OK;
  const int L = cov.n_leaves;
  int nd_max = 0;
  for (int l = 0; l < L; ++l) nd_max = cov.leaves[l].ndims > nd_max ? cov.leaves[l].ndims : nd_max;
  const int64_t ldq = ((m + 15) / 16) * 16, ldc = ((nd_max + 1 + 15) / 16) * 16;
  const int64_t chunk = (n < 32768) ? n : 32768;
  double *norms = nullptr, *Q = nullptr, *T = nullptr, *cext = nullptr;
  MLN_TRY(mln_scratch(ctx, sizeof(double) * (size_t)L * (size_t)(n + m), (void**)&norms));
  double* xx = norms;                        // [L][n]
  double* yy = norms + (size_t)L * n;        // [L][m]
  MLN_TRY(sqnorms(ctx, cov, x, n, d, xx));
  MLN_TRY(sqnorms(ctx, cov, c, m, d, yy));
  const int64_t leaf_stride = chunk * ldq;
  MLN_HIP(ctx, mln_dmalloc((void**)&Q, sizeof(double) * (size_t)L * leaf_stride));
  MLN_HIP(ctx, mln_dmalloc((void**)&T, sizeof(double) * (size_t)chunk * ldc));
  MLN_HIP(ctx, mln_dmalloc((void**)&cext, sizeof(double) * (size_t)L * m * ldc));
  int rc = MLN_OK;
  for (int l = 0; l < L; ++l)
    hipLaunchKernelGGL(k_grad_centres_leaf, dim3((unsigned)((m * ldc + 255) / 256)), dim3(256), 0, ctx->stream, cov, l, c,
                       m, d, cext + (size_t)l * m * ldc, ldc);
  MLN_HIP(ctx, hipMemsetAsync(out, 0, sizeof(double) * (size_t)n * d, ctx->stream));
  const int64_t tiles_n = (ldq + TN - 1) / TN;
  for (int64_t r0 = 0; r0 < n && rc == MLN_OK; r0 += chunk) {
    const int64_t rows = (n - r0 < chunk) ? (n - r0) : chunk;
    const int64_t nblk = tiles_n * ((rows + TM - 1) / TM);
    hipLaunchKernelGGL(k_grad_coeff_multi, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, cov, x + r0 * d, rows, c, m,
                       d, xx + r0, n, yy, w, Q, ldq, leaf_stride, tiles_n);
    for (int l = 0; l < L && rc == MLN_OK; ++l) {
      const int nd = cov.leaves[l].ndims;
      GemmArgs g{};
      g.A = Q + (size_t)l * leaf_stride; g.lda = ldq; g.ta = 0; g.B = cext + (size_t)l * m * ldc; g.ldb = ldc; g.tb = 0;
      g.C = T; g.ldc = ldc; g.M = rows; g.N = nd + 1; g.K = m; g.alpha = 1.0; g.beta = 0.0; g.split_k = 1;
      rc = launch_dgemm(ctx, g);
      if (rc != MLN_OK) break;
      hipLaunchKernelGGL(k_grad_combine_leaf, dim3((unsigned)((rows * nd + 255) / 256)), dim3(256), 0, ctx->stream, cov, l,
                         T, ldc, x + r0 * d, rows, d, out + r0 * d);
    }
  }
  hipError_t e = hipGetLastError();
  (void)hipStreamSynchronize(ctx->stream);
  (void)mln_dfree(Q); (void)mln_dfree(T); (void)mln_dfree(cext);
  if (rc == MLN_OK && e != hipSuccess) rc = mln_hip_fail(ctx, e, "predict_gradient_gemm_multi", __FILE__, __LINE__);
  return rc;
}

// Fast predictor gradient for single stationary-leaf kernels (see k_grad_coeff): row chunks of 32768.
int launch_predict_gradient_gemm(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* c,
                                 int64_t m, int d, const double* w, double* out) {
  if (n == 0) return MLN_OK;
  const DevLeaf& lf = cov.leaves[0];
  const int64_t ldq = ((m + 15) / 16) * 16, ldc = ((lf.ndims + 1 + 15) / 16) * 16;
  const int64_t chunk = (n < 32768) ? n : 32768;
  double *norms = nullptr, *Q = nullptr, *T = nullptr, *cext = nullptr;
  MLN_TRY(mln_scratch(ctx, sizeof(double) * (size_t)(n + m), (void**)&norms));
  double* xx = norms;
  double* yy = norms + n;
  MLN_TRY(sqnorms(ctx, cov, x, n, d, xx));
  MLN_TRY(sqnorms(ctx, cov, c, m, d, yy));
  MLN_HIP(ctx, mln_dmalloc((void**)&Q, sizeof(double) * (size_t)chunk * ldq));
  MLN_HIP(ctx, mln_dmalloc((void**)&T, sizeof(double) * (size_t)chunk * ldc));
  MLN_HIP(ctx, mln_dmalloc((void**)&cext, sizeof(double) * (size_t)m * ldc));
  int rc = MLN_OK;
  hipLaunchKernelGGL(k_grad_centres, dim3((unsigned)((m * ldc + 255) / 256)), dim3(256), 0, ctx->stream, cov, c, m, d,
                     cext, ldc);
  const int64_t tiles_n = (ldq + TN - 1) / TN;
  for (int64_t r0 = 0; r0 < n && rc == MLN_OK; r0 += chunk) {
    const int64_t rows = (n - r0 < chunk) ? (n - r0) : chunk;
    const int64_t nblk = tiles_n * ((rows + TM - 1) / TM);
    hipLaunchKernelGGL(k_grad_coeff, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, cov, x + r0 * d, rows, c, m, d,
                       xx + r0, yy, w, Q, ldq, tiles_n);
    GemmArgs g{};
    g.A = Q; g.lda = ldq; g.ta = 0; g.B = cext; g.ldb = ldc; g.tb = 0; g.C = T; g.ldc = ldc;
    g.M = rows; g.N = lf.ndims + 1; g.K = m; g.alpha = 1.0; g.beta = 0.0; g.split_k = 1;
    rc = launch_dgemm(ctx, g);
    if (rc != MLN_OK) break;
    hipLaunchKernelGGL(k_grad_combine, dim3((unsigned)((rows * d + 255) / 256)), dim3(256), 0, ctx->stream, cov, T, ldc,
                       x + r0 * d, rows, d, out + r0 * d);
  }
  hipError_t e = hipGetLastError();
  (void)hipStreamSynchronize(ctx->stream);
  (void)mln_dfree(Q); (void)mln_dfree(T); (void)mln_dfree(cext);
  if (rc == MLN_OK && e != hipSuccess) rc = mln_hip_fail(ctx, e, "predict_gradient_gemm", __FILE__, __LINE__);
  return rc;
}
