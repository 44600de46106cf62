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

// (staging, loop structure and the padded-copy contract on `y`, `yy`: cov_rows_impl.h)
template <int KIND, bool HAS32, int KSTEPS>
__global__ __launch_bounds__(512) void k_kernel_matrix_rows_prod(DevCov cov, const double* __restrict__ x, int64_t n,
                                                                 const double* __restrict__ y, int64_t m, int d,
                                                                 const double* __restrict__ xx,
                                                                 const double* __restrict__ yy,
                                                                 double* __restrict__ out, int64_t ldo, double add_diag,
                                                                 float* __restrict__ out32, int q32) {
  constexpr int YB = TN * NNS + 512;   // one operand buffer: the state tile, then a slot per thread for stores without an element
  __shared__ double ys[2][YB];
  __shared__ double yn[3][512];        // c0^2 |y_state|^2 of the centres ([..][tid < TN] used, the rest absorbs the other threads' stores)
  __shared__ double yt[3][512];        // c1 * their time stamps (same layout)
  constexpr int NST = (TN * (4 * KSTEPS + 1) + 511) / 512;   // TN x d <= TN x (4 KSTEPS + 1) staged values
  // (cov_epilogue.h) scaled squared distances: state leaf from the pre-scaled norms and the MFMA's dot product; time leaf
  // directly from the two PRE-SCALED time stamps, s1 = (c1 t_x - c1 t_c)^2 + c1^2 1e-12 -- the reference's
  // t_x^2 - 2 t_x t_c + t_c^2 + 1e-12 without its cancellation (equal stamps give exactly 1e-12 either way)
  const double c20 = covepi::sq_scale<KIND>(cov.leaves[0]), m20 = -2.0 * c20;
  const double c21 = covepi::sq_scale<KIND>(cov.leaves[1]), c1 = sqrt(c21), eps1 = c21 * 1e-12;
  constexpr int EPI_VALU = (HAS32 ? 41 : 38) * 16 + 12;
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
  double xr[4], xt[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = row0 + lk + 4 * r;
    xr[r] = (row < n) ? c20 * (xx[row] + 1e-12) : 0.0;   // leaf 0's norms come first in the norms buffer
    xt[r] = (row < n) ? c1 * x[row * d + ds] : 0.0;
  }
  double* const out_wg = out + (int64_t)blockIdx.x * 128 * ldo;
  float* const out32_wg = HAS32 ? out32 + (int64_t)blockIdx.x * 128 * ldo : nullptr;
  unsigned lrowb[4], lrowb32[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    lrowb[r] = ((unsigned)(wave * 16 + lk + 4 * r) * (unsigned)ldo + (unsigned)li) * 8u;
    lrowb32[r] = lrowb[r] >> 1;
  }
  for (int e = tid; e < 2 * YB; e += 512) (&ys[0][0])[e] = 0.0;
  __syncthreads();
  const int cnt = TN * d;
  // element e of a tile: centre e / d, column e % d.  State columns go to the MFMA operand tile; the time stamps are
  // fetched once more by the thread that owns the centre's norm (the tile copy of column d - 1 goes to the sink) and
  // take the norms' three-buffer route, scaled by c1.
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
  double sreg[NST], snorm = 0.0, stime = 0.0;
  auto stage_load = [&](int64_t tile) {
    const char* ytile = (const char*)(y + tile * TN * d);
#pragma unroll
    for (int i = 0; i < NST; ++i) sreg[i] = *(const double*)(ytile + goffb[i]);
    stime = *(const double*)(ytile + tb);
    snorm = *(const double*)((const char*)(yy + tile * TN) + nb);
  };
  auto stage_store = [&](int par, int nbuf) {
#pragma unroll
    for (int i = 0; i < NST; ++i) dst[i][par * YB] = sreg[i];
    yn[nbuf][tid] = c20 * snorm;
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
  const int64_t ntiles = (ldo + TN - 1) / TN;
  const bool interior_rows = (int64_t)blockIdx.x * 128 + 128 <= n;
  stage_load(0); stage_store(0, 0);
  if (ntiles > 1) { stage_load(1); stage_store(1, 1); }
  __syncthreads();
  v4d_t accA[4], accB[4];
  mma(0, accA);
  lds_barrier();
  auto step = [&](int64_t t, v4d_t (&cur)[4], v4d_t (&nxt)[4], int par, int ncur, auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const int64_t t2 = FAST ? (t + 2) : ((t + 2 < ntiles) ? (t + 2) : (ntiles - 1));
    stage_load(t2);
    mma(par ^ 1, nxt);
    const int64_t col0 = t * TN;
    if (FAST) {
      char* const ob = (char*)(out_wg + col0);
      char* const ob32 = HAS32 ? (char*)(out32_wg + col0) : nullptr;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const double yc = yn[ncur][16 * tt + li], tc = yt[ncur][16 * tt + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double dt = xt[r] - tc;
          const double v = covepi::leaf_product_s<KIND>(fmax(fma(m20, cur[tt][r], xr[r]) + yc, 1e-300), fma(dt, dt, eps1));
          *(double*)(ob + 128 * tt + lrowb[r]) = v;
          if (HAS32) *(float*)(ob32 + 64 * tt + lrowb32[r]) = surrogate_bits(v, q32);
        }
      }
#pragma unroll
      for (int i = 0; i < 4 * KSTEPS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, EPI_VALU / (4 * KSTEPS), 0);
      }
    } else {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int64_t c = col0 + 16 * tt + li;
        const double yc = yn[ncur][16 * tt + li], tc = yt[ncur][16 * tt + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = row0 + lk + 4 * r;
          if (row < n && c < ldo) {
            const double dt = xt[r] - tc;
            const double v = (c < m) ? covepi::leaf_product_s<KIND>(fmax(fma(m20, cur[tt][r], xr[r]) + yc, 1e-300), fma(dt, dt, eps1))
                                           + ((row == c) ? add_diag : 0.0)
                                     : 0.0;
            out[row * ldo + c] = v;
            if (HAS32) out32[row * ldo + c] = surrogate_bits(v, q32);
          }
        }
      }
    }
    stage_store(par, (ncur == 0) ? 2 : ncur - 1);   // tile t + 2 into the buffers tile t has just released
    lds_barrier();
  };
  const int64_t n_fast = (interior_rows && add_diag == 0.0) ? (m / TN) : 0;
  int64_t t = 0;
  int nc = 0;
  for (; t + 2 <= n_fast; t += 2) {
    step(t, accA, accB, 0, nc, std::true_type{});
    nc = (nc == 2) ? 0 : nc + 1;
    step(t + 1, accB, accA, 1, nc, std::true_type{});
    nc = (nc == 2) ? 0 : nc + 1;
  }
  for (; t < ntiles; ++t) {
    step(t, accA, accB, (int)(t & 1), nc, std::false_type{});
    nc = (nc == 2) ? 0 : nc + 1;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) accA[tt] = accB[tt];
  }
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
