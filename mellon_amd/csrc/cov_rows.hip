// Kernel-matrix pass of the fit: K = cov(x, xu) for one stationary leaf over all d <= 64 columns (reference:
// util.py:351-366 `distance`, cov.py k() of Matern32/52, ExpQuad, Exponential, RatQuad), persistent-row form.
#include "cov_rows.h"

namespace {
using covrows::NNS;
using covrows::TN;

// Kernel matrix, single leaf over all d <= 64 columns, persistent-row form with the matrix pipe and the VALU
// working at the same time: a workgroup of 8 waves owns 128 rows; every wave keeps the MFMA A operands of its
// 16 rows in registers and walks all centre tiles (staged through LDS, three buffers).  In one loop body the
// wave issues the 4 x ksteps MFMAs of tile t+1 into one accumulator set while the sqrt/exp epilogue and the
// stores of tile t run on the other set -- independent instruction streams in one basic block, interleaved
// with sched_group_barrier (1 MFMA : 24 VALU), so neither pipe waits for the other.
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

template <int KIND, bool HAS32, int KSTEPS>
__global__ __launch_bounds__(512) void k_kernel_matrix_rows(DevCov cov, const double* __restrict__ x, int64_t n,
                                                            const double* __restrict__ y, int64_t m, int d,
                                                            const double* __restrict__ xx,
                                                            const double* __restrict__ yy,
                                                            double* __restrict__ out, int64_t ldo, double add_diag,
                                                            float* __restrict__ out32) {
  __shared__ double ys[2][TN * NNS];   // tile t+1 is consumed while tile t+2 lands in the buffer tile t left
  __shared__ double yn[3][TN];
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
  double xr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = row0 + lk + 4 * r;
    xr[r] = (row < n) ? xx[row] : 0.0;
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
    if (tid < TN) yn[nb][tid] = (col0 + tid < m) ? yy[col0 + tid] : 0.0;   // norms: three buffers (the epilogue of tile t reads them one step later)
  };
  auto mma = [&](int buf, v4d_t (&acc)[4]) {
    const double* yb = &ys[buf][li * NNS + lk];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = v4d_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)   // 4 KSTEPS >= d; k columns past d are zero in both operands
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], yb[16 * t * NNS + 4 * ks], acc[t], 0, 0, 0);
  };
  const int64_t ntiles = (ldo + TN - 1) / TN;   // covers the pad columns of the leading dimension
  const bool interior_rows = (int64_t)blockIdx.x * 128 + 128 <= n;
  stage(0);
  if (ntiles > 1) stage(1);
  __syncthreads();
  v4d_t accA[4], accB[4];
  mma(0, accA);
  __syncthreads();
  for (int64_t t = 0; t < ntiles; ++t) {
    const int cur = (int)(t % 3), nxt = (int)((t + 1) & 1);
    if (t + 2 < ntiles) stage(t + 2);                 // into the ys buffer of tile t, whose MFMAs finished last step
    mma(nxt, accB);                                   // tile t + 1 (the last one is a dummy on stale data)
    const int64_t col0 = t * TN;
    if (interior_rows && col0 + TN <= m && add_diag == 0.0) {
      // branch-free epilogue: one basic block together with the MFMAs above
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int64_t c = col0 + 16 * tt + li;
        const double yc = yn[cur][16 * tt + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = row0 + lk + 4 * r;
          const double v = leaf_value_k<KIND>(lf, xr[r], yc, accA[tt][r]);
          out[row * ldo + c] = v;
          if (HAS32) out32[row * ldo + c] = (float)v;
        }
      }
#pragma unroll
      for (int i = 0; i < 4 * KSTEPS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 1280 / (4 * KSTEPS), 0);   // its share of the epilogue VALU
      }
    } else {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int64_t c = col0 + 16 * tt + li;
        const double yc = yn[cur][16 * tt + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = row0 + lk + 4 * r;
          if (row < n && c < ldo) {
            const double v = (c < m) ? leaf_value_k<KIND>(lf, xr[r], yc, accA[tt][r]) + ((row == c) ? add_diag : 0.0) : 0.0;
            out[row * ldo + c] = v;
            if (HAS32) out32[row * ldo + c] = (float)v;
          }
        }
      }
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) accA[tt] = accB[tt];
    __syncthreads();
  }
}

}  // namespace

int launch_kernel_matrix_rows(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m,
                              int d, const double* xx, const double* yy, double* out, int64_t ldo, double add_diag,
                              float* out32) {
  const dim3 grid((unsigned)((n + 127) / 128)), block(512);
#define MLN_KM_ROWS2(KIND, KS)                                                                                       \
  if (out32) hipLaunchKernelGGL((k_kernel_matrix_rows<KIND, true, KS>), grid, block, 0, ctx->stream, cov, x, n, y, m, d, \
                                xx, yy, out, ldo, add_diag, out32);                                                 \
  else hipLaunchKernelGGL((k_kernel_matrix_rows<KIND, false, KS>), grid, block, 0, ctx->stream, cov, x, n, y, m, d, xx, \
                          yy, out, ldo, add_diag, out32);
#define MLN_KM_ROWS(KIND)                                  \
  if (d <= 32) { MLN_KM_ROWS2(KIND, 8) }                   \
  else if (d <= 52) { MLN_KM_ROWS2(KIND, 13) }             \
  else { MLN_KM_ROWS2(KIND, 16) }
  switch (cov.leaves[0].kind) {
    case MLN_K_MATERN32: MLN_KM_ROWS(MLN_K_MATERN32) break;
    case MLN_K_MATERN52: MLN_KM_ROWS(MLN_K_MATERN52) break;
    case MLN_K_EXPQUAD: MLN_KM_ROWS(MLN_K_EXPQUAD) break;
    case MLN_K_EXPONENTIAL: MLN_KM_ROWS(MLN_K_EXPONENTIAL) break;
    default: MLN_KM_ROWS(MLN_K_RATQUAD) break;
  }
#undef MLN_KM_ROWS2
#undef MLN_KM_ROWS
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}
