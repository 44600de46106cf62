// The iterative hot loop: one pass over the cell-sharded factor L (n x m, fp64, row-major) per
// objective evaluation.  Reference: mellon/inference.py:35-92,167-192 (loss), whose autodiff
// gradient is z + L^T (exp(L z + mu + V) - 1); inference.py:291-338 (diagonal Hessian) and
// inference.py:341-354 (f = L z + mu).  The reference reads L twice per evaluation (L z, then
// L^T via reverse mode); here each row block is read ONCE:
//
//   512-thread workgroup; thread t owns the double2 column pairs {t, t+512, ...} (CPT of them,
//   m <= 1024*CPT), keeps z, the gradient partial (and the Hessian-diagonal partial) for those
//   columns in registers, and streams R rows at a time through registers with 16-byte coalesced
//   loads: partial dots -> wave64 shuffle reduction -> 8 wave partials through LDS (fixed order)
//   -> a_i = exp(f_i + V_i) -> g += (a_i - 1) * row, from the SAME registers.  The next R rows
//   are requested before the current ones are consumed (two register sets), so ~2 x R x m x 8 B
//   per CU are in flight against HBM latency with one workgroup per CU.
//
// HBM-bound: algorithmic bytes per launch = n_local * ldl * 8 (+ O(n + n_wg * m)).
// Determinism: rows are split into contiguous per-workgroup ranges; per-workgroup partials are
// summed in workgroup order by k_reduce_obj.
#include <cstdlib>

#include "mln_internal.h"
#include "objective.h"
#include "mln_options.h"

namespace {

constexpr int WG = 512;
constexpr int MLN_FSTAGE = 6144;   // rows of f one workgroup can stage in LDS (48 KB)
enum { MODE_OBJ = 0, MODE_OBJ_HESS = 1, MODE_GEMVT = 2, MODE_FONLY = 3 };

typedef double d2 __attribute__((ext_vector_type(2)));

// One ds_write_b64 the compiler cannot schedule around: written as a plain C++ store, the LDS write of the per-row f
// made the scheduler sink the NEXT row set's global loads below the current set's arithmetic (vmcnt(0) at the loop
// head, +10 % per pass); as an opaque instruction it leaves the load-early order of the source alone.
__device__ __forceinline__ void lds_store_f64(double* base, int idx, double v) {
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)base + 8u * (unsigned)idx;
  asm volatile("ds_write_b64 %0, %1" : : "v"(addr), "v"(v));   // no memory clobber: global loads may move across it
}

// Row subsample (the solver's first phase: the MAP problem of every row_stride-th cell, solver.hip): logical row i of
// the pass is row  r0 + i * rs  of the buffer (and of V, Vdr, weights); rs = 1, r0 = 0 is the whole shard.
struct RowMap { int64_t r0, rs; __device__ __forceinline__ int64_t operator()(int64_t i) const { return r0 + i * rs; } };
__device__ __forceinline__ RowMap row_map(const ObjArgs& a) { return RowMap{a.row_first, a.row_stride > 0 ? a.row_stride : 1}; }

template <int CPT, int R>
__device__ __forceinline__ void load_rows(const d2* __restrict__ L2, int64_t ld2, int64_t row, int64_t row_end,
                                          unsigned tid, d2 (&v)[R][CPT], const RowMap rm, unsigned lim) {
  // Loads are unconditional (no exec-masked branches, so the compiler keeps exact vmcnt counts and the
  // next register set really is in flight while the current one is consumed): rows past the end re-read
  // the first row of the step (their coefficient is 0), lanes past the row end re-read its last pair
  // (their z is 0 and their partial sums are never stored).
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool rok = (row + r) < row_end;
    // wave-uniform row base (scalar registers) + 32-bit per-lane offset -> saddr addressing
    const d2* rowp = L2 + rm(rok ? (row + r) : row) * ld2;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      unsigned off = (unsigned)c * WG + tid;
      off = (off < lim) ? off : lim - 1u;        // lim: pairs left in the row from the pointer (ld2, or less for a column segment)
      v[r][c] = __builtin_nontemporal_load(rowp + off);
    }
  }
}

// V and Vdr of the R rows of a step (clamped to the last row: rows past the end have coefficient 0)
// lds_barrier_unused: the per-step barrier only publishes eight partial dot products through LDS, so
// "s_waitcnt lgkmcnt(0); s_barrier" would do, and it would leave the prefetched rows of the next step in flight where
// __syncthreads() (a full fence, vmcnt(0)) drains them.  Measured at C3: the fp64 kernel is unchanged (5.8 ms) and the
// 32-bit kernel gets SLOWER, 3.10 -> 3.56 ms -- with the drain, the waves of a workgroup re-align at every step and issue
// their next row segments back to back (whole 20 KB rows as one burst); without it they drift apart and the HBM pages
// are revisited.  The full fence stays.

template <int R>
__device__ __forceinline__ void load_lik(const ObjArgs& a, int64_t row, double (&pv)[2][R]) {
  const RowMap rm = row_map(a);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t i = rm((row + r < a.n) ? (row + r) : (a.n - 1));
    pv[0][r] = a.V[i];
    pv[1][r] = a.Vdr[i];
  }
}

// VEC variants (few column pairs per thread: m <= 3072): lane l < R owns row l of the step -- its V / Vdr arrive as ONE
// vector load per lane instead of R wave-uniform ones
template <int R>
__device__ __forceinline__ void load_lik_vec(const ObjArgs& a, int64_t row, int tid, double (&pv)[2][R]) {
  const RowMap rm = row_map(a);
  const int lane = tid & 63;
  const int64_t want = row + (lane < R ? lane : R - 1);
  const int64_t i = rm((want < a.n) ? want : (a.n - 1));
  pv[0][0] = a.V[i];
  pv[1][0] = a.Vdr[i];
}

__device__ __forceinline__ double read_lane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Sum R = 2^K per-lane values over the 64 lanes of a wave with R - 1 + (6 - K) exchanges instead of 6 R: at each of the
// first K steps a lane keeps one half of its values and hands the other half to its partner, so that afterwards lane l
// holds ONE partial sum, of row l >> (6 - K); the remaining 6 - K steps are plain butterflies.  Returns that row's total.
template <int R>
__device__ __forceinline__ double transposed_wave_sum(double (&v)[R], int lane) {
  constexpr int K = (R == 16) ? 4 : (R == 8) ? 3 : (R == 4) ? 2 : (R == 2) ? 1 : 0;
  static_assert((1 << K) == R, "R must be a power of two <= 16");
#pragma unroll
  for (int st = 0; st < K; ++st) {
    const int half = R >> (st + 1), mask = 32 >> st;
    const bool upper = (lane & mask) != 0;
#pragma unroll
    for (int i = 0; i < half; ++i) {
      const double send = upper ? v[i] : v[i + half];
      const double keep = upper ? v[i + half] : v[i];
      v[i] = keep + __shfl_xor(send, mask, 64);
    }
  }
  double s = v[0];
#pragma unroll
  for (int mask = 32 >> K; mask > 0; mask >>= 1) s += __shfl_xor(s, mask, 64);
  return s;
}

template <int CPT, int R, int MODE, bool KEEP, bool VEC = false>
__device__ __forceinline__ void process_rows(const ObjArgs& a, int64_t row, int64_t row_end, int tid, int par,
                                             const d2 (&v)[R][CPT], const d2 (&z)[CPT], d2 (&g)[CPT],
                                             d2 (&h)[CPT], double& loss, double (*red)[8][R], double* fstage,
                                             int64_t fbase, const double (&pv)[2][R], const double capv, int& any_over) {
  // capv (+inf unless the solver's cap is on): beyond t = capv, e^t is continued by its second-order Taylor polynomial there --
  // e^cap (1 + d + d^2 / 2), d = t - cap: a convex C^2 minorant with curvature bounded by e^cap, identical to e^t wherever no
  // row is above the cap (solver.hip "capped start"; round 5: quadratic instead of linear, and in the fp64 / subsample passes too)
  double coef[R], aexp[R];
  const RowMap rm = row_map(a);
  if constexpr (MODE == MODE_GEMVT) {
#pragma unroll
    for (int r = 0; r < R; ++r) coef[r] = (row + r < row_end) ? a.weights[rm(row + r)] : 0.0;
  } else if constexpr (VEC && (MODE == MODE_OBJ || MODE == MODE_OBJ_HESS)) {
    // With one to three column pairs per thread the per-row work that does not shrink with m -- six exchanges per row for
    // the wave's dot product, an exp evaluated by EVERY thread for EVERY row -- outweighs the FMAs (C4, m = 2000: 5.0 TB/s;
    // C2, m = 1000: less).  Here the R dot products are reduced together (transposed_wave_sum), and after the barrier
    // lane l < R alone finishes row l: cross-wave sum, f, exp, likelihood term; the coefficients go back to all lanes
    // through scalar registers (v_readlane).  One exp sequence per step instead of R.
    double dot[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        s = fma(v[r][c].x, z[c].x, s);
        s = fma(v[r][c].y, z[c].y, s);
      }
      dot[r] = s;
    }
    const int wave = tid >> 6, lane = tid & 63;
    constexpr int GRP = 64 / R;                       // lanes that end up with the same row's total
    const double tot = transposed_wave_sum<R>(dot, lane);
    if ((lane & (GRP - 1)) == 0) red[par][wave][lane / GRP] = tot;
    __syncthreads();   // full fence on purpose: see the note at lds_barrier_unused
    const int rl = lane < R ? lane : R - 1;
    double sr = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) sr += red[par][w][rl];
    const bool rok = lane < R && (row + rl) < row_end;
    const double f = sr + a.mu;
    const double Vi = pv[0][0];                       // (this lane's row: load_lik_vec)
    const double tt = f + Vi;
    const bool over = tt > capv;
    const double ex = rok ? exp(over ? capv : tt) : 0.0;
    const double dc = over ? (tt - capv) : 0.0;
    const double e = ex * fma(dc, fma(0.5, dc, 1.0), 1.0);
    const double cf = rok ? (over ? fma(ex, dc, ex - 1.0) : ex - 1.0) : 0.0;
    if (wave == 0 && rok) loss -= (f + pv[1][0]) - e;   // inference.py:89-91 (summed over lanes at the end)
    any_over += (wave == 0 && rok && over) ? 1 : 0;
    if (KEEP) lds_store_f64(fstage, (wave == 0 && rok) ? (int)(row + rl - fbase) : (MLN_FSTAGE + tid), f);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      coef[r] = read_lane_f64(cf, r);
      aexp[r] = (MODE == MODE_OBJ_HESS) ? read_lane_f64(ex, r) : 0.0;
    }
  } else {
    double dot[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        s = fma(v[r][c].x, z[c].x, s);
        s = fma(v[r][c].y, z[c].y, s);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      dot[r] = s;
    }
    const int wave = tid >> 6;
    if ((tid & 63) == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) red[par][wave][r] = dot[r];
    }
    __syncthreads();   // full fence on purpose: see the note at lds_barrier_unused
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += red[par][w][r];
      const bool rok = (row + r) < row_end;
      const double f = s + a.mu;
      if (MODE == MODE_FONLY) {
        // (f_accum: this launch covers one column segment of a wide matrix and adds to what the previous ones left)
        if (rok && tid == 0) a.f_out[row + r] = a.f_accum ? (s + a.f_out[row + r]) : f;
        coef[r] = 0.0;
        aexp[r] = 0.0;
      } else {
        // KEEP: V / Vdr of the row were requested together with the row itself (load_lik) -- a load issued HERE would
        // sit behind the next set's row loads in the in-order return queue (s_waitcnt vmcnt(0))
        const double Vi = KEEP ? pv[0][r] : (rok ? a.V[rm(row + r)] : 0.0);
        const double tt = f + Vi;
        const bool over = tt > capv;
        const double ex = rok ? exp(over ? capv : tt) : 0.0;
        const double dc = over ? (tt - capv) : 0.0;
        const double e = ex * fma(dc, fma(0.5, dc, 1.0), 1.0);
        aexp[r] = ex;
        coef[r] = rok ? (over ? fma(ex, dc, ex - 1.0) : ex - 1.0) : 0.0;
        if (KEEP) { if (tid == 0 && rok) loss -= (f + pv[1][r]) - e; }
        else if (tid == 0 && rok) loss -= (f + a.Vdr[rm(row + r)]) - e;   // inference.py:89-91
        any_over += (tid == 0 && rok && over) ? 1 : 0;
        // f of the row goes to LDS.  Unconditional store by every lane (thread 0 to the row's slot, the others to a
        // per-lane dummy slot): a store under `if (tid == 0)` is a real branch in the loop body, and a global store
        // there serialises the load pipeline -- both cost 10-20 % of the pass.
        if (KEEP) lds_store_f64(fstage, (tid == 0 && rok) ? (int)(row + r - fbase) : (MLN_FSTAGE + tid), f);
      }
    }
  }
  if (MODE != MODE_FONLY) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        g[c].x = fma(coef[r], v[r][c].x, g[c].x);
        g[c].y = fma(coef[r], v[r][c].y, g[c].y);
        if (MODE == MODE_OBJ_HESS) {
          h[c].x = fma(aexp[r] * v[r][c].x, v[r][c].x, h[c].x);
          h[c].y = fma(aexp[r] * v[r][c].y, v[r][c].y, h[c].y);
        }
      }
    }
  }
}

template <int CPT, int R, int MODE, bool KEEP = false, bool VEC = false>
__global__ __launch_bounds__(WG) void k_objective(ObjArgs a) {
  if (a.gate && (*a.gate & 3) != a.gate_want) return;   // uniform: the device-resident solver chose the other copy / is done
  if (a.gate2 && *a.gate2 != a.gate2_want) return;      //          ... or another subsample level
  __shared__ double red[2][8][R];
  const int tid = threadIdx.x;
  const int64_t ld2 = a.ldl / 2;
  const d2* __restrict__ L2 = reinterpret_cast<const d2*>(a.L);
  // contiguous row range of this workgroup, in units of R rows
  const int64_t nsteps = (a.n + R - 1) / R;
  const int64_t per = (nsteps + a.n_wg - 1) / a.n_wg;
  const int64_t s_beg = (int64_t)blockIdx.x * per;
  int64_t s_end = s_beg + per;
  if (s_end > nsteps) s_end = nsteps;

  d2 z[CPT], g[CPT], h[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int64_t col = 2 * ((int64_t)c * WG + tid);
    z[c] = (d2){0.0, 0.0};
    if (MODE != MODE_GEMVT) {
      if (col < a.m) z[c].x = a.z[col];
      if (col + 1 < a.m) z[c].y = a.z[col + 1];
    }
    g[c] = (d2){0.0, 0.0};
    h[c] = (d2){0.0, 0.0};
  }
  double loss = 0.0;
  int any_over = 0;
  const double capv = a.cap ? *a.cap : __builtin_inf();
  // f of this workgroup's rows is staged in LDS and written out once at the end
  __shared__ double fstage[KEEP ? (MLN_FSTAGE + WG) : 1];
  const int64_t fbase = s_beg * R;
  d2 va[R][CPT], vb[R][CPT];
  // Software pipeline with UNCONDITIONAL loads (steps past the end re-read the last step): the load
  // count between any use and the loads it depends on is then static, so the compiler waits with
  // vmcnt(loads of one set) instead of vmcnt(0) and one set is always in flight behind the one consumed.
  const int64_t s_last = s_end - 1;
  double pa[2][R], pb[2][R];
  const RowMap rm = row_map(a);
  const unsigned lim = a.seg_cols > 0 ? (unsigned)(a.seg_left / 2) : (unsigned)ld2;
  auto lik = [&](int64_t r0, double (&pv)[2][R]) { if (VEC) load_lik_vec<R>(a, r0, tid, pv); else load_lik<R>(a, r0, pv); };
  constexpr bool PRE = KEEP || (VEC && (MODE == MODE_OBJ || MODE == MODE_OBJ_HESS));   // V / Vdr travel with the rows (a per-lane load after the barrier would drain the prefetch)
  if (s_beg < s_end) { load_rows<CPT, R>(L2, ld2, s_beg * R, a.n, tid, va, rm, lim); if (PRE) lik(s_beg * R, pa); }
  for (int64_t s = s_beg; s < s_end; s += 2) {
    const int64_t s1 = (s + 1 < s_end) ? s + 1 : s_last, s2 = (s + 2 < s_end) ? s + 2 : s_last;
    load_rows<CPT, R>(L2, ld2, s1 * R, a.n, tid, vb, rm, lim);
    if (PRE) lik(s1 * R, pb);
    process_rows<CPT, R, MODE, KEEP, VEC>(a, s * R, a.n, tid, 0, va, z, g, h, loss, red, fstage, fbase, pa, capv, any_over);
    load_rows<CPT, R>(L2, ld2, s2 * R, a.n, tid, va, rm, lim);
    if (PRE) lik(s2 * R, pa);
    if (s + 1 < s_end) process_rows<CPT, R, MODE, KEEP, VEC>(a, (s + 1) * R, a.n, tid, 1, vb, z, g, h, loss, red, fstage, fbase, pb, capv, any_over);
  }
  if (any_over && a.over_flag) atomicAdd(a.over_flag, any_over);      // (integer count: the order of the adds does not matter)
  if (KEEP) {
    __syncthreads();
    double* fo = a.f_keep[1 - *a.f_slot];
    const int64_t r_beg = s_beg * R, r_end = (s_end * R < a.n) ? s_end * R : a.n;
    for (int64_t i = r_beg + tid; i < r_end; i += WG) fo[i] = fstage[i - r_beg];
  }
  if (MODE != MODE_FONLY) {
    double* pg = a.part_grad + (int64_t)blockIdx.x * a.m_pad;
    const int64_t store_lim = a.seg_cols > 0 ? ((a.seg_cols + 1) & ~(int64_t)1) : a.m_pad;   // a segment only writes its own columns
    const double osc = a.out_scale != 0.0 ? a.out_scale : 1.0;   // subsample passes: the sums stand for row_stride x as many cells
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int64_t col = 2 * ((int64_t)c * WG + tid);
      if (col < store_lim) *reinterpret_cast<d2*>(pg + col) = (d2){g[c].x * osc, g[c].y * osc};
    }
    if (MODE == MODE_OBJ_HESS) {
      double* ph = a.part_hess + (int64_t)blockIdx.x * a.m_pad;
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const int64_t col = 2 * ((int64_t)c * WG + tid);
        if (col < a.m_pad) *reinterpret_cast<d2*>(ph + col) = h[c];
      }
    }
    if (VEC) {      // the likelihood terms sit in lanes 0 .. R-1 of wave 0
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) loss += __shfl_xor(loss, off, 64);
    }
    if (tid == 0 && a.part_loss) a.part_loss[blockIdx.x] = loss * osc;
  }
}

// ---- fp32 copy of the streamed buffer: warm-up passes of the MAP solve ------------------------------------
// Same structure as k_objective<MODE_OBJ>, but a lane load is 16 bytes = 4 consecutive fp32 columns
// (thread t owns the column quads {t, t+512, ...}); rows stay in registers as floats and are widened
// where they are used; every sum is fp64.  Half the HBM bytes per pass.
typedef float f4 __attribute__((ext_vector_type(4)));

template <int CQ, int R, int NW>
__device__ __forceinline__ void load_rows32(const f4* __restrict__ L4, int64_t ld4, int64_t row, int64_t row_end,
                                            unsigned tid, f4 (&v)[R][CQ], const RowMap rm) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool rok = (row + r) < row_end;
    const f4* rowp = L4 + rm(rok ? (row + r) : row) * ld4;
#pragma unroll
    for (int c = 0; c < CQ; ++c) {   // unconditional, clamped: see load_rows
      unsigned off = (unsigned)c * (64 * NW) + tid;
      off = (off < (unsigned)ld4) ? off : (unsigned)ld4 - 1u;
      v[r][c] = __builtin_nontemporal_load(rowp + off);
    }
  }
}

// element of the 32-bit copy as a double: an fp32 value, or (FIXED) the integer numerator of a 32-bit fixed-point
// number -- its 2^-32 is folded into z on the way in and into the gradient partials on the way out, both exact
// (tried: the integer as the mantissa of 2^52 + u, minus 2^52 -- one full-rate v_add_f64 in place of v_cvt_f64_u32:
//  3.20 instead of 3.10 ms per pass; the conversion is not what bounds this kernel)
template <bool FIXED>
__device__ __forceinline__ double elem32(float v) {
  return FIXED ? (double)__float_as_uint(v) : (double)v;
}

template <int CQ, int R, bool GEMVT, bool KEEP, int NW, bool FIXED, bool VEC = false>
__device__ __forceinline__ void process_rows32(const ObjArgs& a, int64_t row, int64_t row_end, int tid, int par,
                                               const f4 (&v)[R][CQ], const double (&z)[CQ][4], double (&g)[CQ][4],
                                               double& loss, double (*red)[NW][R], double* fstage, int64_t fbase,
                                               const double (&pv)[2][R], double capv, double ecap, int& any_over) {
  double coef[R], dot[R];
  const RowMap rm = row_map(a);
  if (GEMVT) {   // grad_j = sum_i weights_i L_ij  (Ridge right-hand side): no row dots, no barrier
#pragma unroll
    for (int r = 0; r < R; ++r) coef[r] = (row + r < row_end) ? a.weights[rm(row + r)] : 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int c = 0; c < CQ; ++c) {
        g[c][0] = fma(coef[r], elem32<FIXED>(v[r][c].x), g[c][0]);
        g[c][1] = fma(coef[r], elem32<FIXED>(v[r][c].y), g[c][1]);
        g[c][2] = fma(coef[r], elem32<FIXED>(v[r][c].z), g[c][2]);
        g[c][3] = fma(coef[r], elem32<FIXED>(v[r][c].w), g[c][3]);
      }
    }
    return;
  }
  if constexpr (VEC) {   // one or two column quads per thread: rows reduced together, one lane per row (see process_rows)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < CQ; ++c) {
        s = fma(elem32<FIXED>(v[r][c].x), z[c][0], s);
        s = fma(elem32<FIXED>(v[r][c].y), z[c][1], s);
        s = fma(elem32<FIXED>(v[r][c].z), z[c][2], s);
        s = fma(elem32<FIXED>(v[r][c].w), z[c][3], s);
      }
      dot[r] = s;
    }
    const int wave = tid >> 6, lane = tid & 63;
    constexpr int GRP = 64 / R;
    const double tot = transposed_wave_sum<R>(dot, lane);
    if ((lane & (GRP - 1)) == 0) red[par][wave][lane / GRP] = tot;
    __syncthreads();   // full fence on purpose: see the note at lds_barrier_unused
    const int rl = lane < R ? lane : R - 1;
    double sr = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) sr += red[par][w][rl];
    const bool rok = lane < R && (row + rl) < row_end;
    const double f = sr + a.mu;
    const double Vi = pv[0][0];
    const double tt = f + Vi;
    const bool over = tt > capv;
    const double ex = exp(over ? capv : tt);
    const double dc = over ? (tt - capv) : 0.0;
    const double e = rok ? ex * fma(dc, fma(0.5, dc, 1.0), 1.0) : 0.0;
    const double cf = rok ? (over ? fma(ex, dc, ex - 1.0) : ex - 1.0) : 0.0;
    if (wave == 0 && rok) loss -= (f + pv[1][0]) - e;
    any_over += (wave == 0 && rok && over) ? 1 : 0;
    if (KEEP) lds_store_f64(fstage, (wave == 0 && rok) ? (int)(row + rl - fbase) : (MLN_FSTAGE + tid), f);
#pragma unroll
    for (int r = 0; r < R; ++r) coef[r] = read_lane_f64(cf, r);
  } else {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < CQ; ++c) {
      s = fma(elem32<FIXED>(v[r][c].x), z[c][0], s);
      s = fma(elem32<FIXED>(v[r][c].y), z[c][1], s);
      s = fma(elem32<FIXED>(v[r][c].z), z[c][2], s);
      s = fma(elem32<FIXED>(v[r][c].w), z[c][3], s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    dot[r] = s;
  }
  const int wave = tid >> 6;
  if ((tid & 63) == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) red[par][wave][r] = dot[r];
  }
  __syncthreads();   // full fence on purpose: see the note at lds_barrier_unused
#pragma unroll
  for (int r = 0; r < R; ++r) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[par][w][r];
    const bool rok = (row + r) < row_end;
    const double f = s + a.mu;
    const double Vi = KEEP ? pv[0][r] : (rok ? a.V[rm(row + r)] : 0.0);
    const double tt = f + Vi;
    const bool over = tt > capv;                                    // capv = +inf unless the solver's cap is on
    const double ex = exp(over ? capv : tt);
    const double dc = over ? (tt - capv) : 0.0;
    const double e = rok ? ex * fma(dc, fma(0.5, dc, 1.0), 1.0) : 0.0;
    coef[r] = rok ? (over ? fma(ex, dc, ex - 1.0) : ex - 1.0) : 0.0;                    // d/dt: e^t below the cap, e^cap (1 + d) above it
    if (KEEP) { if (tid == 0 && rok) loss -= (f + pv[1][r]) - e; }
    else if (tid == 0 && rok) loss -= (f + a.Vdr[rm(row + r)]) - e;
    any_over += (tid == 0 && rok && over) ? 1 : 0;
    if (KEEP) lds_store_f64(fstage, (tid == 0 && rok) ? (int)(row + r - fbase) : (MLN_FSTAGE + tid), f);   // see k_objective
  }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int c = 0; c < CQ; ++c) {
      g[c][0] = fma(coef[r], elem32<FIXED>(v[r][c].x), g[c][0]);
      g[c][1] = fma(coef[r], elem32<FIXED>(v[r][c].y), g[c][1]);
      g[c][2] = fma(coef[r], elem32<FIXED>(v[r][c].z), g[c][2]);
      g[c][3] = fma(coef[r], elem32<FIXED>(v[r][c].w), g[c][3]);
    }
  }
}

// NW waves per workgroup: a row of ld4 column quads is spread over 64 NW CQ lane slots, and NW is chosen so that few
// of them are idle (m = 5000: 1252 quads on 7 x 64 x 3 = 1344 slots, 93 %; 8 waves would use 81 % of 1536)
template <int CQ, int R, bool GEMVT = false, bool KEEP = false, int NW = 8, bool FIXED = false, bool VEC = false>
__global__ __launch_bounds__(64 * NW) void k_objective32(ObjArgs a) {
  if (a.gate && (*a.gate & 3) != a.gate_want) return;
  if (a.gate2 && *a.gate2 != a.gate2_want) return;
  constexpr int WG = 64 * NW;
  __shared__ double red[2][NW][R];
  const int tid = threadIdx.x;
  const int64_t ld4 = a.ldl / 4;
  const f4* __restrict__ L4 = reinterpret_cast<const f4*>(a.L32);
  const int64_t nsteps = (a.n + R - 1) / R;
  const int64_t per = (nsteps + a.n_wg - 1) / a.n_wg;
  const int64_t s_beg = (int64_t)blockIdx.x * per;
  int64_t s_end = s_beg + per;
  if (s_end > nsteps) s_end = nsteps;
  double z[CQ][4], g[CQ][4];
#pragma unroll
  for (int c = 0; c < CQ; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t col = 4 * ((int64_t)c * WG + tid) + e;
      z[c][e] = (!GEMVT && col < a.m) ? a.z[col] * (FIXED ? 0x1p-32 : 1.0) : 0.0;
      g[c][e] = 0.0;
    }
  double loss = 0.0;
  int any_over = 0;
  const double capv = a.cap ? *a.cap : __builtin_inf();
  const double ecap = exp(capv);
  __shared__ double fstage[KEEP ? (MLN_FSTAGE + WG) : 1];
  const int64_t fbase = s_beg * R;
  f4 va[R][CQ], vb[R][CQ];
  const int64_t s_last = s_end - 1;
  double pa[2][R], pb[2][R];
  const RowMap rm = row_map(a);
  constexpr bool PRE = KEEP || (VEC && !GEMVT);
  auto lik = [&](int64_t r0, double (&pv)[2][R]) { if (VEC) load_lik_vec<R>(a, r0, tid, pv); else load_lik<R>(a, r0, pv); };
  if (s_beg < s_end) { load_rows32<CQ, R, NW>(L4, ld4, s_beg * R, a.n, tid, va, rm); if (PRE) lik(s_beg * R, pa); }
  for (int64_t s = s_beg; s < s_end; s += 2) {   // unconditional loads: see k_objective
    const int64_t s1 = (s + 1 < s_end) ? s + 1 : s_last, s2 = (s + 2 < s_end) ? s + 2 : s_last;
    load_rows32<CQ, R, NW>(L4, ld4, s1 * R, a.n, tid, vb, rm);
    if (PRE) lik(s1 * R, pb);
    process_rows32<CQ, R, GEMVT, KEEP, NW, FIXED, VEC>(a, s * R, a.n, tid, 0, va, z, g, loss, red, fstage, fbase, pa, capv, ecap, any_over);
    load_rows32<CQ, R, NW>(L4, ld4, s2 * R, a.n, tid, va, rm);
    if (PRE) lik(s2 * R, pa);
    if (s + 1 < s_end) process_rows32<CQ, R, GEMVT, KEEP, NW, FIXED, VEC>(a, (s + 1) * R, a.n, tid, 1, vb, z, g, loss, red, fstage, fbase, pb, capv, ecap, any_over);
  }
  if (any_over && a.over_flag) atomicAdd(a.over_flag, any_over);      // (integer count: the order of the adds does not matter)
  if (KEEP) {
    __syncthreads();
    double* fo = a.f_keep[1 - *a.f_slot];
    const int64_t r_beg = s_beg * R, r_end = (s_end * R < a.n) ? s_end * R : a.n;
    for (int64_t i = r_beg + tid; i < r_end; i += WG) fo[i] = fstage[i - r_beg];
  }
  double* pg = a.part_grad + (int64_t)blockIdx.x * a.m_pad;
#pragma unroll
  for (int c = 0; c < CQ; ++c) {
    const int64_t col = 4 * ((int64_t)c * WG + tid);
    if (col < a.m_pad) {   // m_pad is a multiple of 16
      const double sc = (FIXED ? 0x1p-32 : 1.0) * (a.out_scale != 0.0 ? a.out_scale : 1.0);
      *reinterpret_cast<d2*>(pg + col) = (d2){g[c][0] * sc, g[c][1] * sc};
      *reinterpret_cast<d2*>(pg + col + 2) = (d2){g[c][2] * sc, g[c][3] * sc};
    }
  }
  if (VEC) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) loss += __shfl_xor(loss, off, 64);
  }
  if (tid == 0 && a.part_loss) a.part_loss[blockIdx.x] = loss * (a.out_scale != 0.0 ? a.out_scale : 1.0);
}

template <int CQ, int R, int NW, bool VEC = false>
int launch_f32_nw(mln_ctx* ctx, const ObjArgs& a) {
  const dim3 grid((unsigned)a.n_wg), block(64 * NW);
  // (never the f-keeping variant: what the 32-bit copy yields is not the final log-density, and its variant of the
  //  loop measured 3.58 instead of 3.23 ms per pass)
  if (a.l32_fixed) {
    if (a.weights) hipLaunchKernelGGL((k_objective32<CQ, R, true, false, NW, true>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((k_objective32<CQ, R, false, false, NW, true, VEC>), grid, block, 0, ctx->stream, a);
  } else {
    if (a.weights) hipLaunchKernelGGL((k_objective32<CQ, R, true, false, NW>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((k_objective32<CQ, R, false, false, NW, false, VEC>), grid, block, 0, ctx->stream, a);
  }
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

template <int CQ, int R, bool VEC = false>
int launch_f32(mln_ctx* ctx, const ObjArgs& a) {
  const int64_t ld4 = a.ldl / 4;
  int nw = (int)((ld4 + 64 * CQ - 1) / (64 * CQ));
  if (nw <= 6) return launch_f32_nw<CQ, R, 6, VEC>(ctx, a);
  if (nw == 7) return launch_f32_nw<CQ, R, 7, VEC>(ctx, a);
  return launch_f32_nw<CQ, R, 8, VEC>(ctx, a);
}

__global__ void k_to_f32(const double* __restrict__ src, float* __restrict__ dst, int64_t count) {
  const int64_t i = 2 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
  if (i + 1 < count) {
    const d2 v = __builtin_nontemporal_load(reinterpret_cast<const d2*>(src + i));
    dst[i] = (float)v.x;
    dst[i + 1] = (float)v.y;
  } else if (i < count) {
    dst[i] = (float)src[i];
  }
}

// out_loss[0] = sum_wg loss ; out_grad[j] = sum_wg grad[wg][j] ; out_grad[m + j] = sum_wg hess[wg][j]
// 16 columns x 16 groups of workgroup-partials per block: each group sums its (n_wg / 16) partials in ascending
// workgroup order, the 16 group sums are added in fixed order -> bit-reproducible.  (64 columns x 4 groups, i.e. 64
// dependent loads per thread, took 30 us between two objective passes; this shape takes a few.)
__global__ __launch_bounds__(256) void k_reduce_obj(ObjArgs a, double* __restrict__ out_loss,
                                                    double* __restrict__ out_grad, int with_hess) {
  if (a.gate && (*a.gate & 3) == MLN_GATE_DONE) return;   // DONE, or PAUSE (the host rebuilds the preconditioner)
  __shared__ double red[2][16][16];
  const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int64_t j = (int64_t)blockIdx.x * 16 + c;
  double s = 0.0, t = 0.0;
  if (j < a.m) {
    const int per = (a.n_wg + 15) / 16;
    const int w0 = grp * per, w1 = (w0 + per < a.n_wg) ? (w0 + per) : a.n_wg;
#pragma unroll 4
    for (int w = w0; w < w1; ++w) s += a.part_grad[(int64_t)w * a.m_pad + j];
    if (with_hess)
      for (int w = w0; w < w1; ++w) t += a.part_hess[(int64_t)w * a.m_pad + j];
  }
  red[0][grp][c] = s;
  red[1][grp][c] = t;
  __syncthreads();
  if (grp == 0 && j < a.m) {
    double gs = 0.0, hs = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { gs += red[0][q][c]; hs += red[1][q][c]; }
    out_grad[j] = gs;
    if (with_hess) out_grad[a.m + j] = hs;
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {   // the loss partials: one wave, fixed shuffle tree
    double l = 0.0;
    if (a.part_loss)
      for (int w = threadIdx.x; w < a.n_wg; w += 64) l += a.part_loss[w];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
    if (threadIdx.x == 0) {
      out_loss[0] = l;
      if (a.over_flag) { out_loss[1] = (double)*a.over_flag; *a.over_flag = 0; }   // (rows above the solver's cap: see ObjArgs)
    }
  }
}

// y[r] = sum_j M[r][j] x[j] for a row-major M (rows x cols, leading dimension ld, all even and 16-byte
// aligned): one wave per row, 16-byte loads, wave64 shuffle reduction.  Used for the m x m (and stacked
// 2m x m / m x 2m) preconditioner products of every objective evaluation: no partial buffers, no
// separate reduction launch.
__global__ __launch_bounds__(256) void k_gemv_rows(const double* __restrict__ M, int64_t ld, int64_t rows,
                                                   int64_t cols, const double* __restrict__ x,
                                                   double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const d2* __restrict__ row = reinterpret_cast<const d2*>(M + r * ld);
  const d2* __restrict__ xv = reinterpret_cast<const d2*>(x);
  const int64_t pairs = cols / 2;
  double s0 = 0.0, s1 = 0.0;
  for (int64_t p = lane; p < pairs; p += 64) {
    const d2 a = row[p], b = xv[p];
    s0 = fma(a.x, b.x, s0);
    s1 = fma(a.y, b.y, s1);
  }
  double s = s0 + s1;
  if ((cols & 1) && lane == 0) s = fma(M[r * ld + cols - 1], x[cols - 1], s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) y[r] = s;
}

// The same product for the stacked preconditioner operators, whose m x m blocks are triangular: only the non-zero
// part of each row is read (half the bytes).  upper: row i of every block of `blk` rows holds columns [i, ncol);
// lower: columns [0, i] -- in each of the (1 or 2) column segments, the second one starting at column `mseg` of M
// and element `xseg` of x.  Rows past the first block go to y2 when it is given.
__global__ __launch_bounds__(256) void k_gemv_rows_tri(GemvTri g) {
  if (g.gate && (*g.gate & 3) == MLN_GATE_DONE) return;
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= g.rows) return;
  const int64_t i = r % g.blk;
  const int64_t c_lo = (g.upper == 1) ? (i & ~(int64_t)1) : 0, c_hi = g.upper ? g.ncol : i + 1;
  const double* __restrict__ rowp = g.M + r * g.ld;
  double s0 = 0.0, s1 = 0.0;
  const int nseg = g.mseg ? 2 : 1;
  for (int sg = 0; sg < nseg; ++sg) {
    const double* __restrict__ rb = rowp + sg * g.mseg;
    const double* __restrict__ xb = g.x + sg * g.xseg;
    const d2* __restrict__ row = reinterpret_cast<const d2*>(rb);
    const d2* __restrict__ xv = reinterpret_cast<const d2*>(xb);
    // x + xadd in the first segment (the implicit mode's last product: R^-1 (Kj w + K^T (a - 1)))
    const d2* __restrict__ xa = (g.xadd && sg == 0) ? reinterpret_cast<const d2*>(g.xadd) : nullptr;
    // four 16-byte loads per lane in flight (a single one per iteration left the row at ~2 TB/s: latency-bound)
    int64_t p = c_lo / 2 + lane;
    const int64_t p_end = c_hi / 2;            // pairs [p, p_end) are complete
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0, t4 = 0.0, t5 = 0.0;
    for (; p + 192 < p_end; p += 256) {
      const d2 a0 = row[p], a1 = row[p + 64], a2 = row[p + 128], a3 = row[p + 192];
      d2 b0 = xv[p], b1 = xv[p + 64], b2 = xv[p + 128], b3 = xv[p + 192];
      if (xa) {
        const d2 c0 = xa[p], c1 = xa[p + 64], c2 = xa[p + 128], c3 = xa[p + 192];
        b0.x += c0.x; b0.y += c0.y; b1.x += c1.x; b1.y += c1.y; b2.x += c2.x; b2.y += c2.y; b3.x += c3.x; b3.y += c3.y;
      }
      s0 = fma(a0.x, b0.x, s0); s1 = fma(a0.y, b0.y, s1);
      t0 = fma(a1.x, b1.x, t0); t1 = fma(a1.y, b1.y, t1);
      t2 = fma(a2.x, b2.x, t2); t3 = fma(a2.y, b2.y, t3);
      t4 = fma(a3.x, b3.x, t4); t5 = fma(a3.y, b3.y, t5);
    }
    for (; p < p_end; p += 64) {
      const d2 a = row[p];
      d2 b = xv[p];
      if (xa) { const d2 c = xa[p]; b.x += c.x; b.y += c.y; }
      s0 = fma(a.x, b.x, s0);
      s1 = fma(a.y, b.y, s1);
    }
    s0 += (t0 + t2) + t4;
    s1 += (t1 + t3) + t5;
    if ((c_hi & 1) && lane == 0) s0 = fma(rb[c_hi - 1], xb[c_hi - 1] + (xa ? g.xadd[c_hi - 1] : 0.0), s0);
  }
  double s = s0 + s1;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) {
    if (g.y2 && r >= g.blk) g.y2[r - g.blk] = s;
    else g.y[r] = s;
  }
}

template <int CPT, int R, bool VEC = false>
int launch_mode(mln_ctx* ctx, const ObjArgs& a, int mode) {
  dim3 grid((unsigned)a.n_wg), block(WG);
  switch (mode) {
    case MODE_OBJ:
      if (a.f_slot) hipLaunchKernelGGL((k_objective<CPT, R, MODE_OBJ, true, VEC>), grid, block, 0, ctx->stream, a);
      else hipLaunchKernelGGL((k_objective<CPT, R, MODE_OBJ, false, VEC>), grid, block, 0, ctx->stream, a);
      break;
    case MODE_OBJ_HESS: hipLaunchKernelGGL((k_objective<CPT, R, MODE_OBJ_HESS, false, VEC>), grid, block, 0, ctx->stream, a); break;
    case MODE_GEMVT: hipLaunchKernelGGL((k_objective<CPT, R, MODE_GEMVT>), grid, block, 0, ctx->stream, a); break;
    default: hipLaunchKernelGGL((k_objective<CPT, R, MODE_FONLY>), grid, block, 0, ctx->stream, a); break;
  }
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

}  // namespace

int launch_to_f32(mln_ctx* ctx, const double* src, float* dst, int64_t count) {
  if (count <= 0) return MLN_OK;
  const int64_t blocks = (count / 2 + 256) / 256;
  hipLaunchKernelGGL(k_to_f32, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, src, dst, count);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int objective_max_m() { return 65535; }          // beyond 8192 columns: the segmented two-pass route below
static constexpr int64_t MLN_SEG = 8192;        // columns one workgroup can own in registers (CPT <= 8)

namespace {

// per row: a = exp(f + V), coefficient a - 1 of the gradient, and the row's likelihood term; per-block loss partials
__global__ __launch_bounds__(256) void k_lik_rows(const double* __restrict__ f, const double* __restrict__ V,
                                                  const double* __restrict__ Vdr, int64_t n, double* __restrict__ coef,
                                                  double* __restrict__ part_loss, int n_part) {
  __shared__ double red[4];
  const int64_t per = (n + n_part - 1) / n_part;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = (r0 + per < n) ? r0 + per : n;
  double l = 0.0;
  for (int64_t i = r0 + threadIdx.x; i < r1; i += 256) {
    const double e = exp(f[i] + V[i]);
    coef[i] = e - 1.0;
    l -= (f[i] + Vdr[i]) - e;                   // inference.py:89-91
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
  __syncthreads();
  if (threadIdx.x == 0) part_loss[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

int launch_objective(mln_ctx* ctx, const ObjArgs& a);

// m > 8192 columns: no workgroup can hold a whole row's coefficients in registers, so the pass is split the way the
// reference evaluates it -- f = L z over column segments, the row-wise likelihood, then L^T (a - 1) over the same
// segments: two reads of the buffer instead of one.  Host-driven evaluations only (no device-resident solver, no
// 32-bit copy, no Hessian diagonal).
static int launch_objective_wide(mln_ctx* ctx, const ObjArgs& a) {
  if (a.gate || a.part_hess || a.L32 || (a.row_stride > 1)) {
    mln_set_error(ctx, "objective: this mode is not available beyond 8192 landmarks");
    return MLN_ERR_UNSUPPORTED;
  }
  const bool gemvt = a.weights != nullptr, fonly = !gemvt && a.f_out != nullptr;
  double *ftmp = nullptr, *coef = nullptr;
  const int64_t n1 = a.n > 0 ? a.n : 1;
  if (!gemvt && !fonly) {
    MLN_HIP(ctx, mln_dmalloc((void**)&ftmp, sizeof(double) * (size_t)n1 * 2));
    coef = ftmp + n1;
  }
  int rc = MLN_OK;
  if (!gemvt) {                                    // f = L z + mu, segment by segment
    for (int64_t c0 = 0; c0 < a.m && rc == MLN_OK; c0 += MLN_SEG) {
      ObjArgs s = a;
      s.L = a.L + c0; s.m = std::min<int64_t>(MLN_SEG, a.m - c0); s.z = a.z + c0;
      s.ldl = a.ldl; s.m_pad = a.m_pad;
      s.weights = nullptr; s.part_hess = nullptr; s.part_loss = nullptr;
      s.f_out = fonly ? a.f_out : ftmp;
      s.f_accum = c0 > 0 ? 1 : 0;
      s.f_slot = nullptr;
      s.seg_cols = s.m; s.seg_left = a.ldl - c0;
      rc = launch_objective(ctx, s);
    }
  }
  if (rc == MLN_OK && !gemvt && !fonly) {
    hipLaunchKernelGGL(k_lik_rows, dim3((unsigned)a.n_wg), dim3(256), 0, ctx->stream, ftmp, a.V, a.Vdr, a.n, coef,
                       a.part_loss, a.n_wg);
    if (hipGetLastError() != hipSuccess) rc = MLN_ERR_HIP;
  }
  if (rc == MLN_OK && !fonly) {                     // grad = L^T coef, segment by segment
    for (int64_t c0 = 0; c0 < a.m && rc == MLN_OK; c0 += MLN_SEG) {
      ObjArgs s = a;
      s.L = a.L + c0; s.m = std::min<int64_t>(MLN_SEG, a.m - c0);
      s.weights = gemvt ? a.weights : coef;
      s.part_grad = a.part_grad + c0; s.part_hess = nullptr; s.part_loss = nullptr; s.f_out = nullptr; s.f_slot = nullptr;
      s.seg_cols = s.m; s.seg_left = a.ldl - c0;
      rc = launch_objective(ctx, s);
    }
  }
  if (ftmp) { (void)hipStreamSynchronize(ctx->stream); (void)mln_dfree(ftmp); }
  return rc;
}

// rows per workgroup the f-staging of the objective kernels can hold: callers leave f_slot null beyond it
int objective_max_m_one_pass() { return (int)MLN_SEG; }
bool objective_can_keep_f(int64_t n, int n_wg) { return n_wg > 0 && (n + n_wg - 1) / n_wg + 16 <= MLN_FSTAGE; }

int launch_objective(mln_ctx* ctx, const ObjArgs& a) {
  if (a.ldl % 2 != 0) { mln_set_error(ctx, "objective: leading dimension of L must be even"); return MLN_ERR_ARG; }
  if (a.seg_cols == 0 && a.m > MLN_SEG) return launch_objective_wide(ctx, a);
  int mode = MODE_OBJ;
  if (a.weights) mode = MODE_GEMVT;
  else if (a.f_out) mode = MODE_FONLY;
  else if (a.part_hess) mode = MODE_OBJ_HESS;
  if (a.L32 && (mode == MODE_OBJ || mode == MODE_GEMVT) && a.ldl % 4 == 0) {   // fp32 copy: 4 columns per 16-byte lane load
    const int cq = (int)((a.ldl / 4 + WG - 1) / WG);
    const bool vec32 = mode == MODE_OBJ;
    switch (cq) {
      case 1: return vec32 ? launch_f32<1, 8, true>(ctx, a) : launch_f32<1, 8>(ctx, a);     // (R = 16 / 8 spill: the widening to
      case 2: return vec32 ? launch_f32<2, 4, true>(ctx, a) : launch_f32<2, 4>(ctx, a);     //  fp64 doubles the registers per element)
      case 3:
        // row-per-lane likelihood (see process_rows): m = 5000, 7 waves: <3, 2, VEC> 2.93 ms per pass = 0.855 of the HBM
        // spec against 3.19 ms = 0.786 for the plain <3, 3>; <3, 4, VEC> 3.16 ms
        return vec32 ? launch_f32<3, 2, true>(ctx, a) : launch_f32<3, 3>(ctx, a);
      default: return launch_f32<4, 2>(ctx, a);
    }
  }
  // (a column segment of a wide matrix: the workgroup owns seg_cols columns, the row pitch stays ldl)
  const int64_t pairs = a.seg_cols > 0 ? (a.seg_cols + 1) / 2 : a.ldl / 2;
  const int cpt = (int)((pairs + WG - 1) / WG);
  // rows per step chosen so that one register set holds <= 12 double2 per thread
  // (few column pairs per thread: the row-per-lane likelihood with more rows per barrier, see process_rows)
  const bool vec = mode == MODE_OBJ || mode == MODE_OBJ_HESS;
  switch (cpt) {
    case 1: return vec ? launch_mode<1, 8, true>(ctx, a, mode) : launch_mode<1, 8>(ctx, a, mode);     // (R = 16 spills)
    case 2: return vec ? launch_mode<2, 8, true>(ctx, a, mode) : launch_mode<2, 5>(ctx, a, mode);
    case 3: return vec ? launch_mode<3, 4, true>(ctx, a, mode) : launch_mode<3, 3>(ctx, a, mode);
    case 4: return launch_mode<4, 2>(ctx, a, mode);
    case 5: return launch_mode<5, 2>(ctx, a, mode);   // (tried: <5, 2, VEC> 5.90 ms, <5, 4, VEC> 6.00 ms against 5.87 -- the plain variant already streams at the rate of a plain read)
    case 6: return launch_mode<6, 2>(ctx, a, mode);
    case 7:
    case 8: return launch_mode<8, 1>(ctx, a, mode);
    default:
      mln_set_error(ctx, "objective: m > 8192 landmarks is not supported by this build");
      return MLN_ERR_UNSUPPORTED;
  }
}

int launch_gemv_rows(mln_ctx* ctx, const double* M, int64_t ld, int64_t rows, int64_t cols, const double* x,
                     double* y) {
  if (rows <= 0) return MLN_OK;
  if ((ld & 1) || ((uintptr_t)M & 15) || ((uintptr_t)x & 15)) { mln_set_error(ctx, "gemv_rows: unaligned operands"); return MLN_ERR_ARG; }
  hipLaunchKernelGGL(k_gemv_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, M, ld, rows, cols, x, y);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_gemv_tri(mln_ctx* ctx, const GemvTri& g) {
  if (g.rows <= 0) return MLN_OK;
  if ((g.ld & 1) || (g.mseg & 1) || (g.xseg & 1) || ((uintptr_t)g.M & 15) || ((uintptr_t)g.x & 15)) {
    mln_set_error(ctx, "gemv_rows_tri: unaligned operands");
    return MLN_ERR_ARG;
  }
  hipLaunchKernelGGL(k_gemv_rows_tri, dim3((unsigned)((g.rows + 3) / 4)), dim3(256), 0, ctx->stream, g);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_gemv_rows_tri(mln_ctx* ctx, const double* M, int64_t ld, int64_t rows, const double* x, double* y,
                         int upper, int64_t blk, int64_t ncol, int64_t seg) {
  GemvTri g{M, ld, rows, x, y, nullptr, upper, blk, ncol, seg, seg, nullptr};
  return launch_gemv_tri(ctx, g);
}

int launch_reduce_obj2(mln_ctx* ctx, const ObjArgs& a, double* out_loss, double* out_grad) {
  hipLaunchKernelGGL(k_reduce_obj, dim3((unsigned)((a.m + 15) / 16)), dim3(256), 0, ctx->stream, a, out_loss, out_grad,
                     a.part_hess ? 1 : 0);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_reduce_obj(mln_ctx* ctx, const ObjArgs& a, double* out) { return launch_reduce_obj2(ctx, a, out, out + 1); }
