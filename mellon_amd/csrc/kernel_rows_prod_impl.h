// (included by kernel_rows_prod_*.hip, one translation unit per kernel kind)
// Kernel-matrix pass of the time-sensitive fit: K = k(ls, active_dims = :-1)(x, xu) * k(ls_time, active_dims = -1)(x, xu)
// (reference: parameters.py:641-644 builds the product, inference/conditional code consumes cov(x, xu) like any other),
// in the persistent-row form of cov_rows_impl.h: 8 waves x 16 rows keep the MFMA A operands of their STATE columns in
// registers and walk the centre tiles through LDS; the epilogue evaluates both leaves per element -- the state leaf from
// the MFMA's dot product, the time leaf from the two time stamps -- and stores their product (and its 32-bit fixed-point
// copy).  Until round 3 every 2-leaf program took the LDS-tiled VALU kernel: 33 ms of C4's 72 ms step.
#pragma once
#include "cov_rows_impl.h"
#include "predict_rows_prod.h"

namespace {

template <int KIND, bool HAS32, int KSTEPS>
__global__ __launch_bounds__(512) void k_kernel_matrix_rows_prod(DevCov cov, const double* __restrict__ x, int64_t n,
                                                                 const double* __restrict__ y, int64_t m, int d,
                                                                 const double* __restrict__ xx,
                                                                 const double* __restrict__ yy,
                                                                 double* __restrict__ out, int64_t ldo, double add_diag,
                                                                 float* __restrict__ out32, int q32) {
  __shared__ double ys[2][TN * NNS];
  __shared__ double yn[3][512];        // |y_state|^2 of the centres ([..][tid < TN] used, the rest absorbs the other threads' stores)
  __shared__ double yt[3][TN];         // their time stamps
  __shared__ double sink[512];
  constexpr int NST = (TN * (4 * KSTEPS + 1) + 511) / 512;   // TN x d <= TN x (4 KSTEPS + 1) staged values
  const DevLeaf lf0 = cov.leaves[0], lf1 = cov.leaves[1];
  const int ds = d - 1;                // state columns; column d - 1 is the time stamp
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 16;
  double a[KSTEPS];
  {
    const int64_t ar = (row0 + li < n) ? row0 + li : n - 1;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int k = 4 * ks + lk;
      a[ks] = (k < ds) ? x[ar * d + k] : 0.0;
    }
  }
  double xr[4], xt[4], xt2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = row0 + lk + 4 * r;
    xr[r] = (row < n) ? xx[row] : 0.0;                 // leaf 0's norms come first in the norms buffer
    xt[r] = (row < n) ? x[row * d + ds] : 0.0;
    xt2[r] = xt[r] * xt[r];
  }
  double* const out_wg = out + (int64_t)blockIdx.x * 128 * ldo;
  float* const out32_wg = HAS32 ? out32 + (int64_t)blockIdx.x * 128 * ldo : nullptr;
  unsigned lrow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) lrow[r] = (unsigned)(wave * 16 + lk + 4 * r) * (unsigned)ldo + (unsigned)li;
  for (int e = tid; e < 2 * TN * NNS; e += 512) (&ys[0][0])[e] = 0.0;
  __syncthreads();
  const int cnt = TN * d;
  // element e of a tile: centre e / d, column e % d.  State columns go to the MFMA operand tile, the time column to yt.
  int slot[NST], tslot[NST], goff[NST];
  const int64_t last = m * (int64_t)d - 1;
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int e = tid + 512 * i;
    const int r = e / d, k = e - r * d;
    slot[i] = (e < cnt && k < ds) ? (r * NNS + k) : -1;
    tslot[i] = (e < cnt && k == ds) ? r : -1;
    goff[i] = r * d + k;
  }
  double sreg[NST], snorm = 0.0;
  auto stage_load = [&](int64_t tile) {
    const int64_t base = tile * TN * d;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int64_t g = base + goff[i];
      sreg[i] = y[(g <= last) ? g : last];
    }
    const int64_t c = tile * TN + (tid & (TN - 1));
    snorm = yy[(c < m) ? c : (m - 1)];
  };
  auto stage_store = [&](int64_t tile, int ybuf, int nbuf) {
    double* yb = ys[ybuf];
    double* tb = yt[nbuf];
    const int64_t base = tile * TN * d;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      double* dst = (slot[i] >= 0) ? (yb + slot[i]) : ((tslot[i] >= 0) ? (tb + tslot[i]) : (sink + tid));
      *dst = (base + goff[i] <= last) ? sreg[i] : 0.0;
    }
    const int64_t c = tile * TN + (tid & (TN - 1));
    yn[nbuf][tid] = (c < m) ? snorm : 0.0;
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
  const int64_t ntiles = (ldo + TN - 1) / TN;
  const bool interior_rows = (int64_t)blockIdx.x * 128 + 128 <= n;
  stage_load(0); stage_store(0, 0, 0);
  if (ntiles > 1) { stage_load(1); stage_store(1, 1, 1); }
  __syncthreads();
  v4d_t accA[4], accB[4];
  mma(0, accA);
  lds_barrier();
  auto step = [&](int64_t t, auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const int cur = (int)(t % 3), nxt = (int)((t + 1) & 1);
    const int64_t t2 = (t + 2 < ntiles) ? (t + 2) : (ntiles - 1);
    stage_load(t2);
    mma(nxt, accB);
    const int64_t col0 = t * TN;
    if (FAST) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const double yc = yn[cur][16 * tt + li], tc = yt[cur][16 * tt + li];
        const double tc2 = tc * tc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double v = leaf_value_k<KIND>(lf0, xr[r], yc, accA[tt][r]) * leaf_value_k<KIND>(lf1, xt2[r], tc2, xt[r] * tc);
          const unsigned e = lrow[r] + (unsigned)col0 + 16u * tt;
          out_wg[e] = v;
          if (HAS32) out32_wg[e] = surrogate_bits(v, q32);
        }
      }
#pragma unroll
      for (int i = 0; i < 4 * KSTEPS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 1500 / (4 * KSTEPS), 0);
      }
    } else {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int64_t c = col0 + 16 * tt + li;
        const double yc = yn[cur][16 * tt + li], tc = yt[cur][16 * tt + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = row0 + lk + 4 * r;
          if (row < n && c < ldo) {
            const double v = (c < m) ? leaf_value_k<KIND>(lf0, xr[r], yc, accA[tt][r]) *
                                           leaf_value_k<KIND>(lf1, xt2[r], tc * tc, xt[r] * tc) + ((row == c) ? add_diag : 0.0)
                                     : 0.0;
            out[row * ldo + c] = v;
            if (HAS32) out32[row * ldo + c] = surrogate_bits(v, q32);
          }
        }
      }
    }
    stage_store(t2, (int)(t & 1), (int)((t + 2) % 3));   // into the buffers tile t has just released
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) accA[tt] = accB[tt];
    lds_barrier();
  };
  const int64_t n_fast = (interior_rows && add_diag == 0.0) ? (m / TN) : 0;
  for (int64_t t = 0; t < n_fast; ++t) step(t, std::true_type{});
  for (int64_t t = n_fast; t < ntiles; ++t) step(t, std::false_type{});
}

}  // namespace

template <int KIND>
static int launch_rows_prod_kind(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m, int d,
                                 const double* xx, const double* yy, double* out, int64_t ldo, double add_diag, float* out32,
                                 int q32) {
  const dim3 grid((unsigned)((n + 127) / 128)), block(512);
  const int ds = d - 1;
#define MLN_KM_PROD2(KS)                                                                                                   \
  if (out32) hipLaunchKernelGGL((k_kernel_matrix_rows_prod<KIND, true, KS>), grid, block, 0, ctx->stream, cov, x, n, y, m, d, \
                                xx, yy, out, ldo, add_diag, out32, q32);                                                  \
  else hipLaunchKernelGGL((k_kernel_matrix_rows_prod<KIND, false, KS>), grid, block, 0, ctx->stream, cov, x, n, y, m, d, xx,  \
                          yy, out, ldo, add_diag, out32, q32);
  if (ds <= 32) { MLN_KM_PROD2(8) }
  else if (ds <= 52) { MLN_KM_PROD2(13) }
  else { MLN_KM_PROD2(16) }
#undef MLN_KM_PROD2
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

#define MLN_DEFINE_ROWS_PROD_KIND(NAME, KIND)                                                                            \
  int NAME(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m, int d,               \
           const double* xx, const double* yy, double* out, int64_t ldo, double add_diag, float* out32, int q32) {       \
    return launch_rows_prod_kind<KIND>(ctx, cov, x, n, y, m, d, xx, yy, out, ldo, add_diag, out32, q32);                 \
  }
