// (included by predict_rows_*.hip, one translation unit per kernel kind: the fifteen instantiations in one unit took 25 min)
// Fused predictive mean, persistent-row form (reference: conditional.py:899-906 `_mean`, one output column).
#pragma once
#include "cov_rows.h"
#include "cov_epilogue.h"

namespace {
using covrows::NNS;
using covrows::TN;

// scaled squared distance -> covariance value (cov_epilogue.h); RatQuad, the one kind without an exponential, through pow
template <int KIND>
__device__ __forceinline__ double value_of(double s, double alpha) {
  if (KIND == MLN_K_RATQUAD) return pow(fmax(s, 0.0) + 1.0, -alpha);       // cov.py:453-457, s = (dist / ls)^2 / (2 alpha)
  return covepi::leaf_value_s<KIND>(fmax(s, 1e-300));
}

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Fused predictive mean in the same persistent-row form (conditional.py:899-906, one output): the epilogue of
// tile t multiplies each covariance value by its weight and adds it to the row sums while the MFMAs of tile
// t+1 run; the n' x m matrix never exists.  Round 4: the epilogue of cov_epilogue.h (the library sqrt / exp cost ~100
// instructions per element here) and the staging of the kernel-matrix pass (cov_rows_impl.h): `y`, `yy`, `w` are copies
// padded with 3 TN zero rows, centre tile t+2 is requested into registers at the top of the body and written to LDS at
// its end -- no clamps, no masks, constant addresses -- behind an LDS-only barrier; the loop is unrolled by two so that
// the accumulator sets swap roles.
template <int KIND, int KSTEPS>
__global__ __launch_bounds__(512) void k_predict_mean_rows(DevCov cov, const double* __restrict__ x, int64_t n,
                                                           const double* __restrict__ y, int64_t m, int d,
                                                           const double* __restrict__ xx,
                                                           const double* __restrict__ yy,
                                                           const double* __restrict__ w, double mu,
                                                           double* __restrict__ out) {
  constexpr int YB = TN * NNS + 512;
  __shared__ double ys[2][YB];
  __shared__ double yn[3][512];        // scaled norms of the centres ([..][tid < TN] used; the rest absorbs the other threads' stores)
  __shared__ double yw[3][512];        // their weights (0 behind the last centre: masks the pad columns)
  constexpr int NST = (TN * 4 * KSTEPS + 511) / 512;
  constexpr int EPI_VALU = (KIND == MLN_K_RATQUAD ? 120 : 30) * 16 + 8;
  const double c2 = covepi::sq_scale<KIND>(cov.leaves[0]), m2 = -2.0 * c2, alpha = cov.leaves[0].alpha;
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
    xr[r] = (row < n) ? c2 * (xx[row] + 1e-12) : 0.0;
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
    dst[i] = (e < cnt) ? &ys[0][r * NNS + k] : &ys[0][TN * NNS + tid];
  }
  const unsigned nb = (unsigned)(tid & (TN - 1)) * 8u;
  double sreg[NST], snorm = 0.0, sw = 0.0;
  auto stage_load = [&](int64_t tile) {
    const char* yt = (const char*)(y + tile * TN * d);
#pragma unroll
    for (int i = 0; i < NST; ++i) sreg[i] = *(const double*)(yt + goffb[i]);
    snorm = *(const double*)((const char*)(yy + tile * TN) + nb);
    sw = *(const double*)((const char*)(w + tile * TN) + nb);
  };
  auto stage_store = [&](int par, int nbuf) {
#pragma unroll
    for (int i = 0; i < NST; ++i) dst[i][par * YB] = sreg[i];
    yn[nbuf][tid] = c2 * snorm;
    yw[nbuf][tid] = sw;
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
  stage_load(1); stage_store(1, 1);                                   // (tile 1 exists at least as padding)
  __syncthreads();
  v4d_t accA[4], accB[4];
  mma(0, accA);
  lds_barrier();
  auto step = [&](int64_t t, v4d_t (&cur)[4], v4d_t (&nxt)[4], int par, int ncur) {
    stage_load(t + 2);                                                // within the padding: (ntiles + 2) TN < m + 3 TN
    mma(par ^ 1, nxt);                                                // tile t + 1 (the last one works on padding)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const double yc = yn[ncur][16 * tt + li], wc = yw[ncur][16 * tt + li];
#pragma unroll
      for (int r = 0; r < 4; ++r) part[r] = fma(value_of<KIND>(fma(m2, cur[tt][r], xr[r]) + yc, alpha), wc, part[r]);
    }
    if (KIND != MLN_K_RATQUAD) {                                      // (the library pow: left to the scheduler's own devices)
#pragma unroll
      for (int i = 0; i < 4 * KSTEPS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, EPI_VALU / (4 * KSTEPS), 0);
      }
    }
    stage_store(par, (ncur == 0) ? 2 : ncur - 1);                     // tile t + 2 into the buffers tile t has just released
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
  if (t < ntiles) step(t, accA, accB, 0, nc);                         // odd tile count (t is even here)
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
static int launch_predict_rows_kind(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m,
                                    int d, const double* xx, const double* yy, const double* w, double mu, double* out) {
  const dim3 grid((unsigned)((n + 127) / 128)), block(512);
#define MLN_PM_ROWS2(KS) \
  hipLaunchKernelGGL((k_predict_mean_rows<KIND, KS>), grid, block, 0, ctx->stream, cov, x, n, y, m, d, xx, yy, w, mu, out);
  if (d <= 32) { MLN_PM_ROWS2(8) }
  else if (d <= 52) { MLN_PM_ROWS2(13) }
  else { MLN_PM_ROWS2(16) }
#undef MLN_PM_ROWS2
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

#define MLN_DEFINE_PREDICT_ROWS_KIND(NAME, KIND)                                                                        \
  int NAME(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m, int d,              \
           const double* xx, const double* yy, const double* w, double mu, double* out) {                               \
    return launch_predict_rows_kind<KIND>(ctx, cov, x, n, y, m, d, xx, yy, w, mu, out);                                 \
  }
