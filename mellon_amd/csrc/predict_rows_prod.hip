// Fused predictive mean for the time-sensitive product kernel  k(ls, active_dims=:-1) * k(ls_time, active_dims=-1)
// (reference: parameters.py:641-644 builds it, conditional.py:899-906 `_mean` consumes it; PredictorTime.mean,
// base_predictor.py:872-948), in the persistent-row form of predict_rows.hip: 8 waves x 16 query rows keep their MFMA A
// operands (the d - 1 state columns) in registers, candidate tiles go through LDS once per workgroup; the epilogue
// evaluates BOTH leaves per element -- the state leaf from the MFMA's dot product, the time leaf from the two time stamps
// -- multiplies by the weight and sums along the row.  Round 2 sent every 2-leaf program to the LDS-tiled VALU kernel
// (C4 predict: 9.5 M cells/s).
#include "cov_rows.h"
#include "predict_rows_prod.h"

namespace {
using covrows::NNS;
using covrows::TN;

template <int KIND>
__device__ __forceinline__ double leaf_value_p(double inv_ls, double xx, double yy, double xy) {
  const double sq = xx - 2.0 * xy + yy + 1e-12;                       // util.py:362-366
  const double dist = sqrt(fmax(sq, 0.0));
  if (KIND == MLN_K_MATERN32) { const double r = 1.7320508075688772 * dist * inv_ls; return (r + 1.0) * exp(-r); }
  if (KIND == MLN_K_MATERN52) { const double r = 2.23606797749979 * dist * inv_ls; return (r + r * r * 0.3333333333333333 + 1.0) * exp(-r); }
  if (KIND == MLN_K_EXPQUAD) { const double r = dist * inv_ls; return exp(-0.5 * (r * r)); }
  const double r = dist * inv_ls;
  return exp(-0.5 * r);                                               // MLN_K_EXPONENTIAL
}

template <int KIND, int KSTEPS>
__global__ __launch_bounds__(512) void k_predict_mean_rows_prod(DevCov cov, const double* __restrict__ x, int64_t n,
                                                                const double* __restrict__ y, int64_t m, int d,
                                                                const double* __restrict__ xx,
                                                                const double* __restrict__ yy,
                                                                const double* __restrict__ w, double mu,
                                                                double* __restrict__ out) {
  __shared__ double ys[2][TN * NNS];
  __shared__ double yn[3][TN];
  __shared__ double yw[3][TN];
  __shared__ double yt[3][TN];
  const double inv_ls0 = cov.leaves[0].alpha_inv_ls[1], inv_ls1 = cov.leaves[1].alpha_inv_ls[1];
  const int ds = d - 1;                                  // state columns; column d - 1 is the time stamp
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 16;
  double a[16];
  {
    const int64_t ar = (row0 + li < n) ? row0 + li : n - 1;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int k = 4 * ks + lk;
      a[ks] = (k < ds) ? x[ar * d + k] : 0.0;
    }
  }
  double xr[4], xt[4], part[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = row0 + lk + 4 * r;
    xr[r] = (row < n) ? xx[row] : 0.0;                   // |x_state|^2 (leaf 0's norms)
    xt[r] = (row < n) ? x[row * d + ds] : 0.0;
    part[r] = 0.0;
  }
  for (int e = tid; e < 2 * TN * NNS; e += 512) (&ys[0][0])[e] = 0.0;
  __syncthreads();
  auto stage = [&](int64_t tile) {
    const int64_t col0 = tile * TN;
    const int buf = (int)(tile & 1), nb = (int)(tile % 3);
    const int cnt = TN * ds;
    for (int e = tid; e < cnt; e += 512) {
      const int r = e / ds, k = e - r * ds;
      ys[buf][r * NNS + k] = (col0 + r < m) ? y[(col0 + r) * d + k] : 0.0;
    }
    if (tid < TN) {
      const bool ok = col0 + tid < m;
      yn[nb][tid] = ok ? yy[col0 + tid] : 0.0;
      yw[nb][tid] = ok ? w[col0 + tid] : 0.0;            // weight 0 masks the columns past m
      yt[nb][tid] = ok ? y[(col0 + tid) * d + ds] : 0.0;
    }
  };
  auto mma = [&](int buf, v4d_t (&acc)[4]) {
    const double* yb = &ys[buf][li * NNS + lk];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = v4d_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], yb[16 * t * NNS + 4 * ks], acc[t], 0, 0, 0);
  };
  const int64_t ntiles = (m + TN - 1) / TN;
  stage(0);
  if (ntiles > 1) stage(1);
  __syncthreads();
  v4d_t accA[4], accB[4];
  mma(0, accA);
  __syncthreads();
  for (int64_t t = 0; t < ntiles; ++t) {
    const int cur = (int)(t % 3), nxt = (int)((t + 1) & 1);
    if (t + 2 < ntiles) stage(t + 2);
    mma(nxt, accB);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const double yc = yn[cur][16 * tt + li], wc = yw[cur][16 * tt + li], tc = yt[cur][16 * tt + li];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double k0 = leaf_value_p<KIND>(inv_ls0, xr[r], yc, accA[tt][r]);
        const double k1 = leaf_value_p<KIND>(inv_ls1, xt[r] * xt[r], tc * tc, xt[r] * tc);
        part[r] = fma(k0 * k1, wc, part[r]);
      }
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) accA[tt] = accB[tt];
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double s_ = part[r];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s_ += __shfl_xor(s_, off, 64);
    const int64_t row = row0 + lk + 4 * r;
    if (li == 0 && row < n) out[row] = mu + s_;
  }
}

}  // namespace

// The program  leaf0 leaf1 MUL  with leaf0 over columns 0 .. d-2, leaf1 over column d-1, both of the same stationary kind
bool predict_rows_prod_eligible(const DevCov& cov, int d) {
  if (cov.n_leaves != 2 || cov.n_toks != 3 || d < 2 || d > 65) return false;
  if (cov.tok_op[0] != MLN_OP_LEAF || cov.tok_op[1] != MLN_OP_LEAF || cov.tok_op[2] != MLN_OP_MUL) return false;
  if (cov.tok_leaf[0] != 0 || cov.tok_leaf[1] != 1) return false;
  const DevLeaf &l0 = cov.leaves[0], &l1 = cov.leaves[1];
  if (l0.kind != l1.kind || l0.kind < MLN_K_MATERN32 || l0.kind > MLN_K_EXPONENTIAL) return false;
  if (l0.ndims != d - 1 || l1.ndims != 1 || cov.dims[l1.dims_off] != d - 1) return false;
  for (int k = 0; k < d - 1; ++k) if (cov.dims[l0.dims_off + k] != k) return false;
  return true;
}

int launch_predict_mean_rows_prod(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m,
                                  int d, const double* xx0, const double* yy0, const double* w, double mu, double* out) {
  const dim3 grid((unsigned)((n + 127) / 128)), block(512);
  const int ds = d - 1;
#define MLN_PP2(KIND, KS) \
  hipLaunchKernelGGL((k_predict_mean_rows_prod<KIND, KS>), grid, block, 0, ctx->stream, cov, x, n, y, m, d, xx0, yy0, w, mu, out);
#define MLN_PP(KIND)                                  \
  if (ds <= 32) { MLN_PP2(KIND, 8) }                  \
  else if (ds <= 52) { MLN_PP2(KIND, 13) }            \
  else { MLN_PP2(KIND, 16) }
  switch (cov.leaves[0].kind) {
    case MLN_K_MATERN32: MLN_PP(MLN_K_MATERN32) break;
    case MLN_K_MATERN52: MLN_PP(MLN_K_MATERN52) break;
    case MLN_K_EXPQUAD: MLN_PP(MLN_K_EXPQUAD) break;
    default: MLN_PP(MLN_K_EXPONENTIAL) break;
  }
#undef MLN_PP
#undef MLN_PP2
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_kernel_matrix_rows_prod(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m,
                                   int d, const double* xx0, const double* yy0, double* out, int64_t ldo, double add_diag,
                                   float* out32, int q32) {
  switch (cov.leaves[0].kind) {
    case MLN_K_MATERN32: return launch_kernel_matrix_rows_prod_matern32(ctx, cov, x, n, y, m, d, xx0, yy0, out, ldo, add_diag, out32, q32);
    case MLN_K_MATERN52: return launch_kernel_matrix_rows_prod_matern52(ctx, cov, x, n, y, m, d, xx0, yy0, out, ldo, add_diag, out32, q32);
    case MLN_K_EXPQUAD: return launch_kernel_matrix_rows_prod_expquad(ctx, cov, x, n, y, m, d, xx0, yy0, out, ldo, add_diag, out32, q32);
    case MLN_K_EXPONENTIAL: return launch_kernel_matrix_rows_prod_exponential(ctx, cov, x, n, y, m, d, xx0, yy0, out, ldo, add_diag, out32, q32);
    default: break;
  }
  mln_set_error(ctx, "kernel_matrix_rows_prod: unsupported kind");
  return MLN_ERR_UNSUPPORTED;
}
