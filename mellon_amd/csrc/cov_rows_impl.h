// (included by cov_rows_k*.hip, one translation unit per kernel kind: 30 instantiations in one unit take ~10 min)
// Kernel-matrix pass of the fit: K = cov(x, xu) for one stationary leaf over all d <= 64 columns (reference:
// util.py:351-366 `distance`, cov.py k() of Matern32/52, ExpQuad, Exponential, RatQuad), persistent-row form.
#include <type_traits>

#pragma once
#include "cov_rows.h"

namespace {
using covrows::NNS;
using covrows::TN;

// Kernel matrix, single leaf over all d <= 64 columns, persistent-row form with the matrix pipe and the VALU
// working at the same time: a workgroup of 8 waves owns 128 rows; every wave keeps the MFMA A operands of its
// 16 rows in registers and walks all centre tiles (staged through LDS).  In one loop body the wave issues the
// 4 x ksteps MFMAs of tile t+1 into one accumulator set while the sqrt/exp epilogue and the stores of tile t run on
// the other set -- independent instruction streams in one basic block, interleaved with sched_group_barrier, so
// neither pipe waits for the other.
//
// Round-2 changes, from the counters in profiles/r02_pmc_sq.txt (72 VALU instructions per element, waves parked in
// s_waitcnt / s_barrier for 50 % of their cycles):
//   * the barrier of the tile loop waits for LDS traffic only (`s_waitcnt lgkmcnt(0)` + `s_barrier`): __syncthreads()
//     is a full fence, i.e. vmcnt(0) -- every wave sat out the acknowledgement of the 32 row-segment stores it had
//     just issued, once per tile;
//   * the centre tile t+2 is REQUESTED at the top of the body (global loads into registers) and written to LDS at the
//     bottom: the wait in between is vmcnt(#stores issued since), not vmcnt(0);
//   * sqrt and exp are straight-line code (v_rsq_f64 + Newton, Cody-Waite reduction + degree-12 polynomial +
//     v_ldexp_f64) instead of the library routines with their special-case selects: <= 2 ulp, ~45 instructions.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// sqrt(s) for s >= 0 (0 maps to ~1e-150, i.e. to covariance 1 like the reference's sqrt(0) = 0)
__device__ __forceinline__ double sqrt_pos(double s) {
  s = fmax(s, 1e-300);
  double y = __builtin_amdgcn_rsq(s);
  const double hs = 0.5 * s;
  y = y * fma(-hs * y, y, 1.5);
  y = y * fma(-hs * y, y, 1.5);
  double r = s * y;
  return fma(0.5 * y, fma(-r, r, s), r);     // one Heron correction: <= 1 ulp
}

// e^x for x <= 0:  x = k ln2 + t, |t| <= ln2 / 2;  e^t by its Taylor polynomial of degree 12 (|error| < 2e-16)
__device__ __forceinline__ double exp_nonpos(double x) {
  const double k = rint(x * 1.4426950408889634);
  double t = fma(k, -6.93147180369123816490e-01, x);
  t = fma(k, -1.90821492927058770002e-10, t);
  double p = 2.08767569878680989792e-09;            // 1/12!
  p = fma(p, t, 2.50521083854417187751e-08);        // 1/11!
  p = fma(p, t, 2.75573192239858906526e-07);        // 1/10!
  p = fma(p, t, 2.75573192239858906526e-06);        // 1/9!
  p = fma(p, t, 2.48015873015873015873e-05);        // 1/8!
  p = fma(p, t, 1.98412698412698412698e-04);        // 1/7!
  p = fma(p, t, 1.38888888888888888889e-03);        // 1/6!
  p = fma(p, t, 8.33333333333333333333e-03);        // 1/5!
  p = fma(p, t, 4.16666666666666666667e-02);        // 1/4!
  p = fma(p, t, 1.66666666666666666667e-01);        // 1/3!
  p = fma(p, t, 0.5);
  p = fma(p, t, 1.0);
  p = fma(p, t, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)fmax(k, -1100.0));   // underflows to 0 like exp()
}

// the 32-bit copy: the fixed-point number round(v 2^32), 1 stored as 2^32 - 1 -- see mln_internal.h.  (The rows kernels
// only ever write this format -- every kind they handle is bounded by 1; an fp32 copy, MELLON_AMD_SURROGATE=float,
// takes the tiled kernel.  A run-time choice between the two cost 5 instead of 3 instructions per element here.)
// No conversion instruction: v + 2^20 has an ulp of 2^-32, so the low word of that double IS round(v 2^32) (to nearest
// even); v is clamped to 1 - 2^-32 first, where the sum would carry into bit 32.
__device__ __forceinline__ float surrogate_bits(double v, int) {
  return __uint_as_float((unsigned)__double2loint(fmin(v, 0x1.fffffffep-1) + 0x1p20));
}

template <int KIND>
__device__ __forceinline__ double leaf_value_k(const DevLeaf& lf, double xx, double yy, double xy) {
  const double inv_ls = lf.alpha_inv_ls[1];
  const double sq = xx - 2.0 * xy + yy + 1e-12;             // util.py:362-366
  const double dist = sqrt_pos(fmax(sq, 0.0));
  if (KIND == MLN_K_MATERN32) { const double r = 1.7320508075688772 * dist * inv_ls; return (r + 1.0) * exp_nonpos(-r); }
  if (KIND == MLN_K_MATERN52) { const double r = 2.23606797749979 * dist * inv_ls; return (r + r * r * 0.3333333333333333 + 1.0) * exp_nonpos(-r); }
  if (KIND == MLN_K_EXPQUAD) { const double r = dist * inv_ls; return exp_nonpos(-0.5 * (r * r)); }
  const double r = dist * inv_ls;                           // MLN_K_EXPONENTIAL
  return exp_nonpos(-0.5 * r);
}

template <int KIND, bool HAS32, int KSTEPS>
__global__ __launch_bounds__(512) void k_kernel_matrix_rows(DevCov cov, const double* __restrict__ x, int64_t n,
                                                            const double* __restrict__ y, int64_t m, int d,
                                                            const double* __restrict__ xx,
                                                            const double* __restrict__ yy,
                                                            double* __restrict__ out, int64_t ldo, double add_diag,
                                                            float* __restrict__ out32, int q32) {
  __shared__ double ys[2][TN * NNS];   // tile t+1 is consumed while tile t+2 lands in the buffer tile t left
  __shared__ double yn[3][512];        // norms of the centres: [..][tid < TN] used, the rest absorbs the other threads' stores
  __shared__ double sink[512];         // where the staging stores of threads without an element go
  constexpr int NST = (TN * 4 * KSTEPS + 511) / 512;   // staging registers per thread: TN x d <= TN x 4 KSTEPS values
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
  // Store addresses of the branch-free epilogue: a wave-uniform base per workgroup (SGPR pair) + a 32-bit element index
  // per lane, shared by the fp64 buffer and its 32-bit copy.  Two sets of four 64-bit lane addresses pushed the variant
  // with the copy over its register budget: the compiler then re-issued the staging loads late, each followed by
  // s_waitcnt vmcnt(0) -- a drain of all 64 stores in flight, seven times per tile (25.3 ms against 21.0 without the copy).
  double* const out_wg = out + (int64_t)blockIdx.x * 128 * ldo;
  float* const out32_wg = HAS32 ? out32 + (int64_t)blockIdx.x * 128 * ldo : nullptr;
  unsigned lrow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) lrow[r] = (unsigned)(wave * 16 + lk + 4 * r) * (unsigned)ldo + (unsigned)li;
  for (int e = tid; e < 2 * TN * NNS; e += 512) (&ys[0][0])[e] = 0.0;
  __syncthreads();
  const int cnt = TN * d;
  // element e of a tile: row e / d, column e % d; its LDS slot never changes from tile to tile.  Loads and stores are
  // UNCONDITIONAL (clamped address / sink slot, value selected afterwards): an exec-masked load or store is a branch
  // with its own s_waitcnt vmcnt(0), i.e. a stall on every store the wave has in flight.
  int slot[NST];
  int goff[NST];
  const int64_t last = m * (int64_t)d - 1;
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int e = tid + 512 * i;
    const int r = e / d, k = e - r * d;
    slot[i] = (e < cnt) ? (r * NNS + k) : -1;
    goff[i] = r * d + k;
  }
  double sreg[NST], snorm = 0.0;
  auto stage_load = [&](int64_t tile) {          // raw values only: nothing here may CONSUME a load (see stage_store)
    const int64_t base = tile * TN * d;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int64_t g = base + goff[i];
      sreg[i] = y[(g <= last) ? g : last];
    }
    const int64_t c = tile * TN + (tid & (TN - 1));
    snorm = yy[(c < m) ? c : (m - 1)];
  };
  auto stage_store = [&](int64_t tile) {         // masks applied here, after the epilogue's stores have been issued
    double* yb = ys[(int)(tile & 1)];
    const int64_t base = tile * TN * d;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      double* dst = (slot[i] >= 0) ? (yb + slot[i]) : (sink + tid);
      *dst = (base + goff[i] <= last) ? sreg[i] : 0.0;
    }
    const int64_t c = tile * TN + (tid & (TN - 1));
    yn[(int)(tile % 3)][tid] = (c < m) ? snorm : 0.0;   // three buffers: the epilogue of tile t reads them one step later
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
  stage_load(0); stage_store(0);
  if (ntiles > 1) { stage_load(1); stage_store(1); }
  __syncthreads();
  v4d_t accA[4], accB[4];
  mma(0, accA);
  lds_barrier();
  // One tile step.  FAST (interior rows, full tile, no diagonal term) is branch-free: the MFMAs of tile t+1, the
  // epilogue of tile t and its 32 stores form one basic block, so the wait before the LDS writes of tile t+2 counts
  // the stores issued since the loads (vmcnt(32)) instead of draining them.  The two variants run in SEPARATE loops: a
  // branch between them inside one loop makes the compiler assume the slow path's (unknown) store count at the join.
  auto step = [&](int64_t t, auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const int cur = (int)(t % 3), nxt = (int)((t + 1) & 1);
    const int64_t t2 = (t + 2 < ntiles) ? (t + 2) : (ntiles - 1);   // unconditional (a clamped re-load near the end)
    stage_load(t2);                                   // requested now, written to LDS after the epilogue
    mma(nxt, accB);                                   // tile t + 1 (the last one is a dummy on stale data)
    const int64_t col0 = t * TN;
    if (FAST) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int64_t c = col0 + 16 * tt + li;
        const double yc = yn[cur][16 * tt + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double v = leaf_value_k<KIND>(lf, xr[r], yc, accA[tt][r]);
          const unsigned e = lrow[r] + (unsigned)col0 + 16u * tt;
          out_wg[e] = v;
          if (HAS32) out32_wg[e] = surrogate_bits(v, q32);
        }
      }
#pragma unroll
      for (int i = 0; i < 4 * KSTEPS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 800 / (4 * KSTEPS), 0);   // its share of the epilogue VALU
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
            if (HAS32) out32[row * ldo + c] = surrogate_bits(v, q32);
          }
        }
      }
    }
    {                                                 // into the ys buffer of tile t, whose MFMAs finished last step
      double* yb = ys[(int)(t & 1)];                  // (t2 clamped: the buffer of a finished tile takes a harmless copy)
      const int64_t base = t2 * TN * d;
#pragma unroll
      for (int i = 0; i < NST; ++i) {
        double* dst = (slot[i] >= 0) ? (yb + slot[i]) : (sink + tid);
        *dst = (base + goff[i] <= last) ? sreg[i] : 0.0;
      }
      const int64_t c = t2 * TN + (tid & (TN - 1));
      yn[(int)((t + 2) % 3)][tid] = (c < m) ? snorm : 0.0;
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) accA[tt] = accB[tt];
    lds_barrier();
  };
  const int64_t n_fast = (interior_rows && add_diag == 0.0) ? (m / TN) : 0;   // full tiles of interior rows
  for (int64_t t = 0; t < n_fast; ++t) step(t, std::true_type{});
  for (int64_t t = n_fast; t < ntiles; ++t) step(t, std::false_type{});
}

}  // namespace


template <int KIND>
static int launch_rows_kind(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m, int d,
                            const double* xx, const double* yy, double* out, int64_t ldo, double add_diag, float* out32,
                            int q32) {
  const dim3 grid((unsigned)((n + 127) / 128)), block(512);
#define MLN_KM_ROWS2(KS)                                                                                              \
  if (out32) hipLaunchKernelGGL((k_kernel_matrix_rows<KIND, true, KS>), grid, block, 0, ctx->stream, cov, x, n, y, m, d, \
                                xx, yy, out, ldo, add_diag, out32, q32);                                             \
  else hipLaunchKernelGGL((k_kernel_matrix_rows<KIND, false, KS>), grid, block, 0, ctx->stream, cov, x, n, y, m, d, xx,  \
                          yy, out, ldo, add_diag, out32, q32);
  if (d <= 32) { MLN_KM_ROWS2(8) }
  else if (d <= 52) { MLN_KM_ROWS2(13) }
  else { MLN_KM_ROWS2(16) }
#undef MLN_KM_ROWS2
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

#define MLN_DEFINE_ROWS_KIND(NAME, KIND)                                                                                 \
  int NAME(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m, int d,               \
           const double* xx, const double* yy, double* out, int64_t ldo, double add_diag, float* out32, int q32) {       \
    return launch_rows_kind<KIND>(ctx, cov, x, n, y, m, d, xx, yy, out, ldo, add_diag, out32, q32);                      \
  }
