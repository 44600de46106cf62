// (included by predict_rows_prod_*.hip, one translation unit per kernel kind)
// Fused predictive mean for the time-sensitive product kernel  k(ls, active_dims=:-1) * k(ls_time, active_dims=-1)
// (reference: parameters.py:641-644 builds it, conditional.py:899-906 `_mean` consumes it; PredictorTime.mean,
// base_predictor.py:872-948), in the persistent-row form of predict_rows_impl.h: 8 waves x 16 query rows keep their MFMA A
// operands (the d - 1 state columns) in registers, candidate tiles go through LDS once per workgroup; the epilogue
// evaluates BOTH leaves per element -- the state leaf from the MFMA's dot product, the time leaf from the two time stamps
// -- multiplies by the weight and sums along the row.
#pragma once
#include "cov_rows.h"
#include "cov_epilogue.h"
#include "predict_rows_prod.h"

namespace {
using covrows::NNS;
using covrows::TN;

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Round 4: one exponential for both leaves (cov_epilogue.h: p0 e^{-r0} p1 e^{-r1} = p0 p1 e^{-(r0 + r1)}), the time leaf's
// squared distance directly from the two pre-scaled stamps, and the register-staged, branch-free, unrolled-by-two tile
// pipeline of the kernel-matrix pass over padded copies of `y`, `yy`, `w` (cov_rows_impl.h, kernel_rows_prod_impl.h) --
// the library sqrt / exp pairs cost ~200 instructions per element here.
template <int KIND, int KSTEPS>
__global__ __launch_bounds__(512) void k_predict_mean_rows_prod(DevCov cov, const double* __restrict__ x, int64_t n,
                                                                const double* __restrict__ y, int64_t m, int d,
                                                                const double* __restrict__ xx,
                                                                const double* __restrict__ yy,
                                                                const double* __restrict__ w, double mu,
                                                                double* __restrict__ out) {
  constexpr int YB = TN * NNS + 512;
  __shared__ double ys[2][YB];
  __shared__ double yn[3][512];        // c0^2 |y_state|^2 ([..][tid < TN] used; the rest absorbs the other threads' stores)
  __shared__ double yw[3][512];        // weights (0 behind the last centre)
  __shared__ double yt[3][512];        // c1 * time stamp
  constexpr int NST = (TN * (4 * KSTEPS + 1) + 511) / 512;
  constexpr int EPI_VALU = 40 * 16 + 12;
  const double c20 = covepi::sq_scale<KIND>(cov.leaves[0]), m20 = -2.0 * c20;
  const double c21 = covepi::sq_scale<KIND>(cov.leaves[1]), c1 = sqrt(c21), eps1 = c21 * 1e-12;
  const int ds = d - 1;                                  // state columns; column d - 1 is the time stamp
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
  double xr[4], xt[4], part[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = row0 + lk + 4 * r;
    xr[r] = (row < n) ? c20 * (xx[row] + 1e-12) : 0.0;   // |x_state|^2 (leaf 0's norms), scaled
    xt[r] = (row < n) ? c1 * x[row * d + ds] : 0.0;
    part[r] = 0.0;
  }
  for (int e = tid; e < 2 * YB; e += 512) (&ys[0][0])[e] = 0.0;
  __syncthreads();
  const int cnt = TN * d;
  unsigned goffb[NST];
  double* dst[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int e = tid + 512 * i;
    const int r = e / d, k = e - r * d;
    goffb[i] = (e < cnt) ? (unsigned)e * 8u : 0u;
    dst[i] = (e < cnt && k < ds) ? &ys[0][r * NNS + k] : &ys[0][TN * NNS + tid];
  }
  const unsigned nb = (unsigned)(tid & (TN - 1)) * 8u;
  const unsigned tb = ((unsigned)(tid & (TN - 1)) * (unsigned)d + (unsigned)ds) * 8u;
  double sreg[NST], snorm = 0.0, sw = 0.0, stime = 0.0;
  auto stage_load = [&](int64_t tile) {
    const char* ytile = (const char*)(y + tile * TN * d);
#pragma unroll
    for (int i = 0; i < NST; ++i) sreg[i] = *(const double*)(ytile + goffb[i]);
    stime = *(const double*)(ytile + tb);
    snorm = *(const double*)((const char*)(yy + tile * TN) + nb);
    sw = *(const double*)((const char*)(w + tile * TN) + nb);
  };
  auto stage_store = [&](int par, int nbuf) {
#pragma unroll
    for (int i = 0; i < NST; ++i) dst[i][par * YB] = sreg[i];
    yn[nbuf][tid] = c20 * snorm;
    yw[nbuf][tid] = sw;
    yt[nbuf][tid] = c1 * stime;
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
  stage_load(0); stage_store(0, 0);
  stage_load(1); stage_store(1, 1);
  __syncthreads();
  v4d_t accA[4], accB[4];
  mma(0, accA);
  lds_barrier();
  auto step = [&](int64_t t, v4d_t (&cur)[4], v4d_t (&nxt)[4], int par, int ncur) {
    stage_load(t + 2);                                   // within the padding: (ntiles + 2) TN < m + 3 TN
    mma(par ^ 1, nxt);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const double yc = yn[ncur][16 * tt + li], wc = yw[ncur][16 * tt + li], tc = yt[ncur][16 * tt + li];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double dt = xt[r] - tc;
        const double v = covepi::leaf_product_s<KIND>(fmax(fma(m20, cur[tt][r], xr[r]) + yc, 1e-300), fma(dt, dt, eps1));
        part[r] = fma(v, wc, part[r]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4 * KSTEPS; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, EPI_VALU / (4 * KSTEPS), 0);
    }
    stage_store(par, (ncur == 0) ? 2 : ncur - 1);
    lds_barrier();
  };
  int64_t t = 0;
  int nc = 0;
  for (; t + 2 <= ntiles; t += 2) {
    step(t, accA, accB, 0, nc);
    nc = (nc == 2) ? 0 : nc + 1;
    step(t + 1, accB, accA, 1, nc);
    nc = (nc == 2) ? 0 : nc + 1;
  }
  if (t < ntiles) step(t, accA, accB, 0, nc);
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

template <int KIND>
static int launch_predict_rows_prod_kind(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y,
                                         int64_t m, int d, const double* xx0, const double* yy0, const double* w, double mu,
                                         double* out) {
  const dim3 grid((unsigned)((n + 127) / 128)), block(512);
  const int ds = d - 1;
#define MLN_PP2(KS) \
  hipLaunchKernelGGL((k_predict_mean_rows_prod<KIND, KS>), grid, block, 0, ctx->stream, cov, x, n, y, m, d, xx0, yy0, w, mu, out);
  if (ds <= 32) { MLN_PP2(8) }
  else if (ds <= 52) { MLN_PP2(13) }
  else { MLN_PP2(16) }
#undef MLN_PP2
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

#define MLN_DEFINE_PREDICT_ROWS_PROD_KIND(NAME, KIND)                                                                   \
  int NAME(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m, int d,              \
           const double* xx0, const double* yy0, const double* w, double mu, double* out) {                             \
    return launch_predict_rows_prod_kind<KIND>(ctx, cov, x, n, y, m, d, xx0, yy0, w, mu, out);                          \
  }
