// Fused predictive mean, persistent-row form (reference: conditional.py:899-906 `_mean`, one output column).
#include "cov_rows.h"

namespace {
using covrows::NNS;
using covrows::TN;

template <int KIND>
__device__ __forceinline__ double leaf_value_k(const DevLeaf& lf, double xx, double yy, double xy) {
  const double inv_ls = lf.alpha_inv_ls[1];
  const double sq = xx - 2.0 * xy + yy + 1e-12;
  const double dist = sqrt(fmax(sq, 0.0));
  if (KIND == MLN_K_MATERN32) { const double r = 1.7320508075688772 * dist * inv_ls; return (r + 1.0) * exp(-r); }
  if (KIND == MLN_K_MATERN52) { const double r = 2.23606797749979 * dist * inv_ls; return (r + r * r * 0.3333333333333333 + 1.0) * exp(-r); }
  if (KIND == MLN_K_EXPQUAD) { const double r = dist * inv_ls; return exp(-0.5 * (r * r)); }
  if (KIND == MLN_K_EXPONENTIAL) { const double r = dist * inv_ls; return exp(-0.5 * r); }
  const double r = dist * inv_ls;
  return pow(r * r / (2.0 * lf.alpha) + 1.0, -lf.alpha);
}

// Fused predictive mean in the same persistent-row form (conditional.py:899-906, one output): the epilogue of
// tile t multiplies each covariance value by its weight and adds it to the row sums while the MFMAs of tile
// t+1 run; the n' x m matrix never exists.
template <int KIND, int KSTEPS>
__global__ __launch_bounds__(512) void k_predict_mean_rows(DevCov cov, const double* __restrict__ x, int64_t n,
                                                           const double* __restrict__ y, int64_t m, int d,
                                                           const double* __restrict__ xx,
                                                           const double* __restrict__ yy,
                                                           const double* __restrict__ w, double mu,
                                                           double* __restrict__ out) {
  __shared__ double ys[2][TN * NNS];
  __shared__ double yn[3][TN];
  __shared__ double yw[3][TN];
  const DevLeaf lf = cov.leaves[0];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 16;
  double a[16];
  {
    const int64_t ar = (row0 + li < n) ? row0 + li : n - 1;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int k = 4 * ks + lk;
      a[ks] = (k < d) ? x[ar * d + k] : 0.0;
    }
  }
  double xr[4], part[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = row0 + lk + 4 * r;
    xr[r] = (row < n) ? xx[row] : 0.0;
    part[r] = 0.0;
  }
  for (int e = tid; e < 2 * TN * NNS; e += 512) (&ys[0][0])[e] = 0.0;
  __syncthreads();
  auto stage = [&](int64_t tile) {
    const int64_t col0 = tile * TN;
    const int buf = (int)(tile & 1), nb = (int)(tile % 3);
    const int cnt = TN * d;
    for (int e = tid; e < cnt; e += 512) {
      const int r = e / d, k = e - r * d;
      ys[buf][r * NNS + k] = (col0 + r < m) ? y[(col0 + r) * d + k] : 0.0;
    }
    if (tid < TN) {
      yn[nb][tid] = (col0 + tid < m) ? yy[col0 + tid] : 0.0;
      yw[nb][tid] = (col0 + tid < m) ? w[col0 + tid] : 0.0;     // weight 0 masks the columns past m
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
      const double yc = yn[cur][16 * tt + li], wc = yw[cur][16 * tt + li];
#pragma unroll
      for (int r = 0; r < 4; ++r) part[r] = fma(leaf_value_k<KIND>(lf, xr[r], yc, accA[tt][r]), wc, part[r]);
    }
#pragma unroll
    for (int i = 0; i < 4 * KSTEPS; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 1100 / (4 * KSTEPS), 0);
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

int launch_predict_mean_rows(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m,
                             int d, const double* xx, const double* yy, const double* w, double mu, double* out) {
  const dim3 grid((unsigned)((n + 127) / 128)), block(512);
#define MLN_PM_ROWS2(KIND, KS) \
  hipLaunchKernelGGL((k_predict_mean_rows<KIND, KS>), grid, block, 0, ctx->stream, cov, x, n, y, m, d, xx, yy, w, mu, out);
#define MLN_PM_ROWS(KIND)                                  \
  if (d <= 32) { MLN_PM_ROWS2(KIND, 8) }                   \
  else if (d <= 52) { MLN_PM_ROWS2(KIND, 13) }             \
  else { MLN_PM_ROWS2(KIND, 16) }
  switch (cov.leaves[0].kind) {
    case MLN_K_MATERN32: MLN_PM_ROWS(MLN_K_MATERN32) break;
    case MLN_K_MATERN52: MLN_PM_ROWS(MLN_K_MATERN52) break;
    case MLN_K_EXPQUAD: MLN_PM_ROWS(MLN_K_EXPQUAD) break;
    case MLN_K_EXPONENTIAL: MLN_PM_ROWS(MLN_K_EXPONENTIAL) break;
    default: MLN_PM_ROWS(MLN_K_RATQUAD) break;
  }
#undef MLN_PM_ROWS
#undef MLN_PM_ROWS2
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}
