// Analytic covariance gradients (reference: Covariance.k_grad of mellon/cov.py:68-100,163-202,261-299,
// 358-396,459-499,558-596 and the Add/Mul/Pow rules of mellon/base_cov.py:317-497; util.distance_grad,
// util.py:369-425) and the gradient of the predictive mean (Predictor.gradient, base_predictor.py:490-505,
// which the reference obtains with jax.jacrev of `_mean`).
//
// A covariance program P(k_1..k_L) over L <= 4 leaves differentiates as
//     d cov / d y = sum_l (dP/dk_l) * g_l(dist_l) * (y - x)|dims_l         (stationary leaves)
//                 + sum_l (dP/dk_l) * x|dims_l / ls_l                      (Linear leaves)
// with the scalar radial factor g_l = phi_l'(dist) / denominator.  dP/dk_l comes from evaluating the
// postfix program once per leaf with (value, tangent) pairs -- a handful of scalar operations next to
// the d-long dot products.  The n x m x d tensor of k_grad exists only because the API returns it;
// the predictor gradient contracts it with the weights on the fly: one thread per query row, the row
// and its gradient accumulator in LDS ([dim][row]: conflict-free), centre tiles broadcast from LDS.
#include <cstdlib>
#include "mln_internal.h"
#include "cov_program.h"

namespace {

constexpr int GR = 128;   // query rows per workgroup (predict gradient)
constexpr int GC = 64;    // centres per LDS tile

// value and radial gradient factor of one leaf.  exact_denominator = 0: distance_grad's
// delta / (dist + 1e-12) (k_grad); 1: the exact derivative of sqrt(max(sq, 0)), delta / dist (what
// autodiff of `_mean` gives), 0 where the clamp is active.
__device__ __forceinline__ void leaf_value_grad(const DevLeaf& lf, double xx, double yy, double xy,
                                                int exact_denominator, double* k, double* g) {
  const double inv_ls = lf.alpha_inv_ls[1];
  if (lf.kind == MLN_K_LINEAR) { *k = xy * inv_ls; *g = inv_ls; return; }
  const double sq = xx - 2.0 * xy + yy + 1e-12;
  const double dist = sqrt(fmax(sq, 0.0));
  const double inv = exact_denominator ? ((sq > 0.0) ? 1.0 / dist : 0.0) : 1.0 / (dist + 1e-12);
  switch (lf.kind) {
    case MLN_K_MATERN32: {                                   // cov.py:84-97
      const double f = 1.7320508075688772 * inv_ls, r = f * dist, e = exp(-r);
      *k = (r + 1.0) * e;
      *g = -f * r * e * inv;
      break;
    }
    case MLN_K_MATERN52: {                                   // cov.py:186-199
      const double f = 2.23606797749979 * inv_ls, r = f * dist, e = exp(-r);
      *k = (r + r * r * 0.3333333333333333 + 1.0) * e;
      *g = -0.3333333333333333 * e * r * (r + 1.0) * f * inv;
      break;
    }
    case MLN_K_EXPQUAD: {                                    // cov.py:283-296
      const double r = dist * inv_ls, e = exp(-0.5 * (r * r));
      *k = e;
      *g = -r * inv_ls * e * inv;
      break;
    }
    case MLN_K_EXPONENTIAL: {                                // cov.py:380-393
      const double r = dist * inv_ls, e = exp(-0.5 * r);
      *k = e;
      *g = -0.5 * inv_ls * e * inv;
      break;
    }
    default: {                                               // RatQuad cov.py:481-496
      const double r = dist * inv_ls, b = r * r / (2.0 * lf.alpha) + 1.0;
      *k = pow(b, -lf.alpha);
      *g = -r * inv_ls * pow(b, -lf.alpha - 1.0) * inv;
    }
  }
}

// out[i][j][:] = d cov(x_i, y_j) / d y_j ; one thread per pair
__global__ void k_kernel_grad(DevCov cov, const double* __restrict__ x, int64_t n, const double* __restrict__ y,
                              int64_t m, int d, int exact_denominator, double* __restrict__ out) {
  const int64_t total = n * m;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx / m, j = idx % m;
    const double* xi = x + i * d;
    const double* yj = y + j * d;
    double kv[MLN_MAX_LEAVES], gf[MLN_MAX_LEAVES], a[MLN_MAX_LEAVES];
#pragma unroll
    for (int l = 0; l < MLN_MAX_LEAVES; ++l) {
      kv[l] = 0.0; gf[l] = 0.0;
      if (l >= cov.n_leaves) continue;
      const DevLeaf lf = cov.leaves[l];
      double xx = 0.0, yy = 0.0, xy = 0.0;
      for (int k = 0; k < lf.ndims; ++k) {
        const int dim = cov.dims[lf.dims_off + k];
        const double a_ = xi[dim], b_ = yj[dim];
        xx = fma(a_, a_, xx); yy = fma(b_, b_, yy); xy = fma(a_, b_, xy);
      }
      leaf_value_grad(lf, xx, yy, xy, exact_denominator, &kv[l], &gf[l]);
    }
    program_adjoints(cov, kv, a);
    double* o = out + idx * d;
    for (int dim = 0; dim < d; ++dim) o[dim] = 0.0;
#pragma unroll
    for (int l = 0; l < MLN_MAX_LEAVES; ++l) {
      if (l >= cov.n_leaves) continue;
      const DevLeaf lf = cov.leaves[l];
      const double c = a[l] * gf[l];
      for (int k = 0; k < lf.ndims; ++k) {
        const int dim = cov.dims[lf.dims_off + k];
        o[dim] += (lf.kind == MLN_K_LINEAR) ? c * xi[dim] : c * (yj[dim] - xi[dim]);
      }
    }
  }
}

// out[i][:] = sum_j w_j d cov(x_i, c_j) / d x_i
__global__ __launch_bounds__(GR) void k_predict_gradient(DevCov cov, const double* __restrict__ x, int64_t n,
                                                          const double* __restrict__ c, int64_t m, int d,
                                                          const double* __restrict__ w, double* __restrict__ out) {
  extern __shared__ double smem[];
  double* xs = smem;                        // [d][GR]
  double* G = xs + (size_t)d * GR;          // [d][GR]
  double* cs = G + (size_t)d * GR;          // [GC][d]
  double* ws = cs + (size_t)GC * d;         // [GC]
  double* yys = ws + GC;                    // [MLN_MAX_LEAVES][GC]
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * GR;
  for (int idx = tid; idx < GR * d; idx += GR) {
    const int r = idx / d, dim = idx % d;
    xs[dim * GR + r] = (row0 + r < n) ? x[(row0 + r) * d + dim] : 0.0;
    G[dim * GR + r] = 0.0;
  }
  __syncthreads();
  double xxl[MLN_MAX_LEAVES];
#pragma unroll
  for (int l = 0; l < MLN_MAX_LEAVES; ++l) {
    xxl[l] = 0.0;
    if (l >= cov.n_leaves) continue;
    const DevLeaf lf = cov.leaves[l];
    for (int k = 0; k < lf.ndims; ++k) {
      const double v = xs[cov.dims[lf.dims_off + k] * GR + tid];
      xxl[l] = fma(v, v, xxl[l]);
    }
  }
  const bool live = row0 + tid < n;
  for (int64_t col0 = 0; col0 < m; col0 += GC) {
    __syncthreads();
    for (int idx = tid; idx < GC * d; idx += GR) {
      const int j = idx / d;
      cs[idx] = (col0 + j < m) ? c[(col0 + j) * d + (idx % d)] : 0.0;
    }
    if (tid < GC) ws[tid] = (col0 + tid < m) ? w[col0 + tid] : 0.0;
    __syncthreads();
    if (tid < GC) {
#pragma unroll
      for (int l = 0; l < MLN_MAX_LEAVES; ++l) {
        if (l >= cov.n_leaves) continue;
        const DevLeaf lf = cov.leaves[l];
        double s = 0.0;
        for (int k = 0; k < lf.ndims; ++k) {
          const double v = cs[tid * d + cov.dims[lf.dims_off + k]];
          s = fma(v, v, s);
        }
        yys[l * GC + tid] = s;
      }
    }
    __syncthreads();
    if (!live) continue;
    const int jn = (int)((m - col0 < GC) ? (m - col0) : GC);
    for (int j = 0; j < jn; ++j) {
      double kv[MLN_MAX_LEAVES], gf[MLN_MAX_LEAVES], a[MLN_MAX_LEAVES];
#pragma unroll
      for (int l = 0; l < MLN_MAX_LEAVES; ++l) {
        kv[l] = 0.0; gf[l] = 0.0;
        if (l >= cov.n_leaves) continue;
        const DevLeaf lf = cov.leaves[l];
        double xy = 0.0;
        for (int k = 0; k < lf.ndims; ++k) {
          const int dim = cov.dims[lf.dims_off + k];
          xy = fma(xs[dim * GR + tid], cs[j * d + dim], xy);
        }
        leaf_value_grad(lf, xxl[l], yys[l * GC + j], xy, 1, &kv[l], &gf[l]);
      }
      program_adjoints(cov, kv, a);
      const double wj = ws[j];
#pragma unroll
      for (int l = 0; l < MLN_MAX_LEAVES; ++l) {
        if (l >= cov.n_leaves) continue;
        const DevLeaf lf = cov.leaves[l];
        const double q = wj * a[l] * gf[l];
        for (int k = 0; k < lf.ndims; ++k) {
          const int dim = cov.dims[lf.dims_off + k];
          const double cv = cs[j * d + dim];
          const double delta = (lf.kind == MLN_K_LINEAR) ? cv : xs[dim * GR + tid] - cv;
          G[dim * GR + tid] = fma(q, delta, G[dim * GR + tid]);
        }
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < GR * d; idx += GR) {
    const int r = idx / d, dim = idx % d;
    if (row0 + r < n) out[(row0 + r) * d + dim] = G[dim * GR + r];
  }
}

}  // namespace

int launch_kernel_grad(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m,
                       int d, int exact_denominator, double* out) {
  if (n == 0 || m == 0) return MLN_OK;
  int64_t blocks = (n * m + 255) / 256;
  if (blocks > 65535 * 16) blocks = 65535 * 16;
  hipLaunchKernelGGL(k_kernel_grad, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, cov, x, n, y, m, d,
                     exact_denominator, out);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_predict_gradient(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* c, int64_t m,
                            int d, const double* w, double* out) {
  if (n == 0) return MLN_OK;
  // single stationary leaf and enough pairs: covariance-tile pass + GEMM on the matrix cores (cov_kernels.hip)
  if (cov.n_toks == 1 && cov.leaves[0].kind != MLN_K_LINEAR && n * m >= (int64_t)1 << 16)
    return launch_predict_gradient_gemm(ctx, cov, x, n, c, m, d, w, out);
  // composite program whose leaves are all stationary: the same idea with one coefficient matrix per leaf
  bool stationary = cov.n_leaves >= 1;
  for (int l = 0; l < cov.n_leaves; ++l) stationary = stationary && cov.leaves[l].kind != MLN_K_LINEAR;
  if (stationary && n * m >= (int64_t)1 << 16)
    return launch_predict_gradient_gemm_multi(ctx, cov, x, n, c, m, d, w, out);
  const size_t lds = sizeof(double) * ((size_t)2 * d * GR + (size_t)GC * d + GC + (size_t)MLN_MAX_LEAVES * GC);
  if (lds > 160 * 1024) { mln_set_error(ctx, "predict_gradient: too many dimensions for the LDS layout"); return MLN_ERR_UNSUPPORTED; }
  MLN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_predict_gradient),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_predict_gradient, dim3((unsigned)((n + GR - 1) / GR)), dim3(GR), lds, ctx->stream, cov, x, n, c,
                     m, d, w, out);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}
