// (included by cov_rows_k*.hip, one translation unit per kernel kind: 30 instantiations in one unit take ~10 min)
// Kernel-matrix pass of the fit: K = cov(x, xu) for one stationary leaf over all d <= 64 columns (reference:
// util.py:351-366 `distance`, cov.py k() of Matern32/52, ExpQuad, Exponential, RatQuad), persistent-row form.
#include <type_traits>

#pragma once
#include "cov_rows.h"
#include "cov_epilogue.h"

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
//   * sqrt and exp are straight-line code instead of the library routines with their special-case selects (round 2: ~45
//     instructions; round 4: 27, cov_epilogue.h -- scaled squared distance, one Newton step, one-FMA range reduction).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// the 32-bit copy: the fixed-point number round(v 2^32), 1 stored as 2^32 - 1 -- see mln_internal.h.  (The rows kernels
// only ever write this format -- every kind they handle is bounded by 1; an fp32 copy, MELLON_AMD_SURROGATE=float,
// takes the tiled kernel.  A run-time choice between the two cost 5 instead of 3 instructions per element here.)
// No conversion instruction: v + 2^20 has an ulp of 2^-32, so the low word of that double IS round(v 2^32) (to nearest
// even); v is clamped to 1 - 2^-32 first, where the sum would carry into bit 32.
__device__ __forceinline__ float surrogate_bits(double v, int) {
  return __uint_as_float((unsigned)__double2loint(fmin(v, 0x1.fffffffep-1) + 0x1p20));
}

// covariance value from the pre-scaled norms (cov_epilogue.h): xs = c2 (|x|^2 + 1e-12), ys = c2 |y|^2, m2 = -2 c2
template <int KIND>
__device__ __forceinline__ double leaf_value_k(double m2, double xs, double ys, double xy) {
  return covepi::leaf_value_s<KIND>(fmax(fma(m2, xy, xs) + ys, 1e-300));
}

// Contract (round 4): `y` and `yy` are PADDED copies of the centres and of their squared norms -- rows m .. m + 3 TN - 1
// exist and are zero (cov_kernels.hip: pad_rows; launch_kernel_matrix takes these kernels only while ldo <= m + 64) -- so the staging needs neither address clamps nor value masks: a
// staged load is `global_load v, v_off32, s[tile base]` with a byte offset that never changes, its LDS destination a
// constant per thread.  The tile loop is unrolled by two, so that the two accumulator sets swap roles instead of being
// copied (16 v_mov_b64 per tile) and every LDS address of the operand tiles is an immediate.
template <int KIND, bool HAS32, int KSTEPS>
__global__ __launch_bounds__(512) void k_kernel_matrix_rows(DevCov cov, const double* __restrict__ x, int64_t n,
                                                            const double* __restrict__ y, int64_t m, int d,
                                                            const double* __restrict__ xx,
                                                            const double* __restrict__ yy,
                                                            double* __restrict__ out, int64_t ldo, double add_diag,
                                                            float* __restrict__ out32, int q32) {
  constexpr int YB = TN * NNS + 512;   // one operand buffer: the tile, then a slot per thread for stores without an element
  __shared__ double ys[2][YB];         // tile t+1 is consumed while tile t+2 lands in the buffer tile t left
  __shared__ double yn[3][512];        // scaled norms of the centres: [..][tid < TN] used, the rest absorbs the other threads' stores
  constexpr int NST = (TN * 4 * KSTEPS + 511) / 512;   // staging registers per thread: TN x d <= TN x 4 KSTEPS values
  constexpr int EPI_VALU = (HAS32 ? 32 : 29) * 16 + 8;   // VALU instructions of one tile's epilogue per lane
  const double c2 = covepi::sq_scale<KIND>(cov.leaves[0]), m2 = -2.0 * c2;
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
    xr[r] = (row < n) ? c2 * (xx[row] + 1e-12) : 0.0;       // util.py:362-366's + 1e-12 rides with the row norm
  }
  // Store addresses of the branch-free epilogue: a wave-uniform base per tile and column group (SGPR pair, scalar
  // arithmetic) + ONE 32-bit byte offset per lane and row, shared by all 16 column positions (and, halved, by the 32-bit
  // copy): no vector instruction per store.
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
  // element e of a tile: centre e / d, column e % d; its global byte offset within a tile and its LDS slot never change
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
  double sreg[NST], snorm = 0.0;
  auto stage_load = [&](int64_t tile) {          // raw values only: nothing here may CONSUME a load
    const char* yt = (const char*)(y + tile * TN * d);
#pragma unroll
    for (int i = 0; i < NST; ++i) sreg[i] = *(const double*)(yt + goffb[i]);
    snorm = *(const double*)((const char*)(yy + tile * TN) + nb);
  };
  auto stage_store = [&](int par, int nbuf) {    // after the epilogue's stores have been issued
#pragma unroll
    for (int i = 0; i < NST; ++i) dst[i][par * YB] = sreg[i];
    yn[nbuf][tid] = c2 * snorm;                  // three buffers: the epilogue of tile t reads them one step later
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
  stage_load(0); stage_store(0, 0);
  if (ntiles > 1) { stage_load(1); stage_store(1, 1); }
  __syncthreads();
  v4d_t accA[4], accB[4];
  mma(0, accA);
  lds_barrier();
  // One tile step: the epilogue of tile t (accumulators `cur`) while the MFMAs of tile t+1 fill `nxt`.  FAST (interior
  // rows, full tile, no diagonal term) is branch-free: the MFMAs, the epilogue and its 16 stores form one basic block, so
  // the wait before the LDS writes of tile t+2 counts the stores issued since the loads (vmcnt(16)) instead of draining
  // them.  PAR = t & 1 as a compile-time constant in the unrolled loop.  The two variants run in SEPARATE loops: a branch
  // between them inside one loop makes the compiler assume the slow path's (unknown) store count at the join.
  auto step = [&](int64_t t, v4d_t (&cur)[4], v4d_t (&nxt)[4], int par, int ncur, auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const int64_t t2 = FAST ? (t + 2) : ((t + 2 < ntiles) ? (t + 2) : (ntiles - 1));   // (a clamped re-load near the end)
    stage_load(t2);                                   // requested now, written to LDS after the epilogue
    mma(par ^ 1, nxt);                                // tile t + 1 (the last one is a dummy on stale data)
    const int64_t col0 = t * TN;
    if (FAST) {
      char* const ob = (char*)(out_wg + col0);
      char* const ob32 = HAS32 ? (char*)(out32_wg + col0) : nullptr;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const double yc = yn[ncur][16 * tt + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double v = leaf_value_k<KIND>(m2, xr[r], yc, cur[tt][r]);
          *(double*)(ob + 128 * tt + lrowb[r]) = v;
          if (HAS32) *(float*)(ob32 + 64 * tt + lrowb32[r]) = surrogate_bits(v, q32);
        }
      }
#pragma unroll
      for (int i = 0; i < 4 * KSTEPS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, EPI_VALU / (4 * KSTEPS), 0);   // its share of the epilogue VALU
      }
    } else {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int64_t c = col0 + 16 * tt + li;
        const double yc = yn[ncur][16 * tt + li];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = row0 + lk + 4 * r;
          if (row < n && c < ldo) {
            const double v = (c < m) ? leaf_value_k<KIND>(m2, xr[r], yc, cur[tt][r]) + ((row == c) ? add_diag : 0.0) : 0.0;
            out[row * ldo + c] = v;
            if (HAS32) out32[row * ldo + c] = surrogate_bits(v, q32);
          }
        }
      }
    }
    stage_store(par, (ncur == 0) ? 2 : ncur - 1);     // tile t + 2: the ys buffer of tile t (its MFMAs finished last step), yn[(t + 2) % 3]
    lds_barrier();
  };
  const int64_t n_fast = (interior_rows && add_diag == 0.0) ? (m / TN) : 0;   // full tiles of interior rows
  int64_t t = 0;
  int nc = 0;                                         // t % 3
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
