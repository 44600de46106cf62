// k-means landmarks on the device (reference: mellon/parameters.py:243-291
// calls sklearn.cluster.k_means(x, m, n_init=1, random_state=...), i.e. k-means++ seeding + Lloyd).
// Same algorithm family, own implementation: the centroids are NOT bit-compatible with sklearn's
// (different RNG stream and summation order), so parity tests pass landmarks explicitly; this is
// the at-scale path (sklearn needs minutes for 1e6 x 50 cells and 5 000 clusters).
//
//   seeding   k-means++ (D^2 sampling): one fused kernel per new centre updates every cell's squared
//             distance to its closest centre and emits 1024-cell block sums; the host draws the next
//             centre from them with a seeded xorshift generator (two tiny downloads per centre).
//   Lloyd     assignment = tiled fp64 distance kernel with running arg-min (the nn_distances tile
//             structure), update = FIXED-POINT integer atomics into m x d sums (round 4c: any summation order gives
//             the same bits -- the centres are bit-reproducible), stop when the summed squared centre shift
//             <= tol * mean feature variance (sklearn's rule) or after max_iter sweeps.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

#include "mln_internal.h"

hipError_t mln_dfree_synced(void* p);   // alloc.hip: release after the caller synchronised the only stream that used p
#include "rowmin_f16.h"
#include "mln_options.h"

namespace {
constexpr int TM = 64, TN = 64, DK = 16, PADT = 4, SBLK = 1024;

__global__ void k_sqnorm_rows(const double* __restrict__ x, int64_t n, int d, double* __restrict__ xx) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int k = 0; k < d; ++k) { double v = x[i * (int64_t)d + k]; s = fma(v, v, s); }
  xx[i] = s;
}

// mind[i] = min(mind[i], |x_i - c|^2) ; bsum[b] = sum of mind over the 1024-cell block b
__global__ __launch_bounds__(256) void k_seed_update(const double* __restrict__ x, int64_t n, int d,
                                                     const double* __restrict__ c, double* __restrict__ mind,
                                                     double* __restrict__ bsum, int first) {
  __shared__ double cs[64];
  __shared__ double red[256];
  for (int k = threadIdx.x; k < d; k += 256) cs[k] = c[k];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SBLK;
  double acc = 0.0;
  for (int q = 0; q < SBLK / 256; ++q) {
    const int64_t i = base + q * 256 + threadIdx.x;
    if (i < n) {
      double s = 0.0;
      for (int k = 0; k < d; ++k) { double t = x[i * (int64_t)d + k] - cs[k]; s = fma(t, t, s); }
      if (!first) s = fmin(s, mind[i]);
      mind[i] = s;
      acc += s;
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[blockIdx.x] = red[0];
}

// index of the first cell of block b whose running sum of mind exceeds `target`
__global__ void k_seed_pick(const double* __restrict__ mind, int64_t n, int64_t b, double target, int64_t* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int64_t lo = b * SBLK, hi = (lo + SBLK < n) ? lo + SBLK : n;
  double run = 0.0;
  int64_t pick = hi - 1;
  for (int64_t i = lo; i < hi; ++i) {
    run += mind[i];
    if (run > target) { pick = i; break; }
  }
  *out = pick;
}

// The same update from the half-precision copy of the cells (rowmin_f16.hip split rows, hi halves: one 128-byte line per
// cell instead of 400 bytes): D^2 only weights the k-means++ draw, three significant digits are plenty, and the 5000
// sequential updates of a 1e6-cell seeding are pure memory traffic.  scale: the copy holds scale * x.
// SB cells per workgroup (and per entry of bsum): 256 -- two passes of 128 cells -- since round 6: the seeding of the first
// level walks 1.7e5 cells, 163 workgroups of 1024 cells left three quarters of the CUs idle (12 us per centre, now see DESIGN.md)
template <int SB>
__global__ __launch_bounds__(1024) void k_seed_update_h(const _Float16* __restrict__ xh, int64_t n, int d, float inv_scale,
                                                       const double* __restrict__ c, const double* __restrict__ prep,
                                                       double* __restrict__ mind, double* __restrict__ bsum, int first) {
  __shared__ float cs[64];
  __shared__ double red[1024];
  // the copy holds -2 (x - centre) * scale (rowmin_prepare): the centre goes through the same map, D^2 comes back in the
  // data's units
  for (int k = threadIdx.x; k < 64; k += 1024) cs[k] = (k < d) ? (float)((c[k] - prep[k]) * prep[64]) : 0.f;
  const double unscale = 1.0 / (prep[64] * prep[64]);
  __syncthreads();
  typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
  // Eight lanes per cell, one 16-byte piece of its 128-byte line each: a wave reads 8 consecutive lines = 1 KB in one
  // instruction.  (One lane per cell -- 64 lanes, 64 different lines per load instruction -- ran at 2.1 TB/s: 61 us per
  // update at 1e6 cells, 0.3 s of a 5000-centre seeding.)
  const int64_t base = (int64_t)blockIdx.x * SB;
  const int sub = threadIdx.x & 7, grp = threadIdx.x >> 3;      // piece of the line, cell within the pass of 128
  const int dk = (d + 7) / 8;
  double acc = 0.0;
  for (int q = 0; q < SB / 128; ++q) {
    const int64_t i = base + q * 128 + grp;
    float s = 0.f;
    if (i < n && sub < dk) {
      const h8_t v = *reinterpret_cast<const h8_t*>(xh + i * 128 + 8 * sub);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 8 * sub + e;
        const float t = (k < d) ? (float)v[e] * inv_scale - cs[k] : 0.f;
        s = fmaf(t, t, s);
      }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (i < n && sub == 0) {
      double sd = (double)s * unscale;
      if (!first) sd = fmin(sd, mind[i]);
      mind[i] = sd;
      acc += sd;
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[blockIdx.x] = red[0];
}

// (Round 6 measured the draw fused into the update, twice: made by the LAST workgroup of the update to finish -- the
// device-scope fence every workgroup then needs took the update from 12 to 54 us; repeated by EVERY workgroup of the next
// update from ping-pong copies of the sums -- same cells drawn, 18.2 us per centre against 9.4 + 4.3: the draw's barriers
// and dependent loads in front of 651 workgroups cost more than a launch.  Two launches it stays.)
// One k-means++ draw on the device (no host round trip per centre): total of the block sums, the block and then the cell
// where the running sum of D^2 passes u * total, and the cell's coordinates copied into the next centre's slot.  u: this
// step's uniform draw (the whole sequence is uploaded once).  Both levels are a 256-wide inclusive scan (wave shuffles + four
// wave totals) and a search for the one thread whose interval [exclusive, inclusive) holds the target -- fixed order, no
// atomics: the same draw in every run.  (Round 3 had thread 0 walk the 256 partial sums, twice: 13 us per draw, 65 ms of a
// 5000-centre seeding, most of it LDS latency.)
__device__ __forceinline__ void seed_scan_find(double v, double target, int t, double* wtot, int* owner, double* rem) {
  const int lane = t & 63, wave = t >> 6;
  double inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  if (lane == 63) wtot[wave] = inc;
  if (t == 0) { *owner = -1; }
  __syncthreads();
  double base = 0.0;
  for (int w = 0; w < wave; ++w) base += wtot[w];
  // exclusive sum = the PREVIOUS lane's inclusive one (the wave's base for lane 0), not incl - v: the intervals then tile
  // [0, total) exactly -- no overlap (two owners writing owner / rem unsynchronised), no gap (owner left at -1)
  const double prev = __shfl_up(inc, 1, 64);
  const double incl = base + inc, excl = (lane == 0) ? base : base + prev;
  // values are non-negative: the inclusive sums are non-decreasing and exactly one thread has excl <= target < incl
  if (v > 0.0 && !(target < excl) && target < incl) { *owner = t; *rem = target - excl; }
  __syncthreads();
}

template <int SB>
__global__ __launch_bounds__(256) void k_seed_select(const double* __restrict__ bsum, int64_t nblk, const double* __restrict__ mind,
                                                     int64_t n, double u, const double* __restrict__ x, int d,
                                                     double* __restrict__ c_next) {
  __shared__ double wtot[4];
  __shared__ int owner;
  __shared__ double rem;
  __shared__ int64_t chosen;
  const int t = threadIdx.x;
  // level 1: which block of SB cells.  Thread t sums its run of block sums; the scan finds the run, its owner the block.
  const int64_t per = (nblk + 255) / 256;
  const int64_t b0 = t * per, b1 = (b0 + per < nblk) ? b0 + per : nblk;
  double ps = 0.0;
  for (int64_t b = b0; b < b1; ++b) ps += bsum[b];
  // total = the last inclusive sum: one more pass of the same scan gives it to everybody
  double inc = ps;
  {
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double o = __shfl_up(inc, off, 64);
      if (lane >= off) inc += o;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
  }
  const double total = ((wtot[0] + wtot[1]) + wtot[2]) + wtot[3];
  __syncthreads();
  if (!(total > 0.0)) {                          // all cells coincide with centres: any cell will do
    if (t == 0) { const int64_t cur = (int64_t)(u * (double)n); chosen = cur >= n ? n - 1 : cur; }
    __syncthreads();
  } else {
    seed_scan_find(ps, u * total, t, wtot, &owner, &rem);
    // (rounding can leave the target at or beyond the last inclusive sum: then the last non-empty run)
    int own = owner;
    double target = rem;
    if (own < 0) { own = 255; while (own > 0 && !((int64_t)own * per < nblk)) --own; target = INFINITY; }
    int64_t blk = (int64_t)own * per;
    {
      const int64_t bend = (blk + per < nblk) ? blk + per : nblk;
      for (; blk + 1 < bend; ++blk) { if (target < bsum[blk]) break; target -= bsum[blk]; }
    }
    __syncthreads();
    // level 2: which cell of the block, the same way (4 cells per thread)
    const int64_t lo = blk * SB, hi = (lo + SB < n) ? lo + SB : n;
    constexpr int CPT = SB / 256;
    double v[CPT], ls = 0.0;
#pragma unroll
    for (int e = 0; e < CPT; ++e) { const int64_t i = lo + t * CPT + e; v[e] = (i < hi) ? mind[i] : 0.0; ls += v[e]; }
    seed_scan_find(ls, target, t, wtot, &owner, &rem);
    if (owner < 0) {                             // target beyond the block's sum (rounding): its last cell that is not a centre yet
      int last = -1;
#pragma unroll
      for (int e = 0; e < CPT; ++e) if (v[e] > 0.0) last = t * CPT + e;
      __syncthreads();
      if (last >= 0) atomicMax(&owner, last);    // (integer maximum: order-independent)
      __syncthreads();
      if (t == 0) chosen = owner >= 0 ? lo + owner : hi - 1;
    } else if (t == owner) {
      double run = 0.0;
      int64_t pick = lo + t * CPT + CPT - 1;
#pragma unroll
      for (int e = 0; e < CPT; ++e) { run += v[e]; if (run > rem) { pick = lo + t * CPT + e; break; } }
      chosen = pick < hi ? pick : hi - 1;
    }
    __syncthreads();
  }
  for (int k = t; k < d; k += 256) c_next[k] = x[chosen * d + k];
}

// label[i] = argmin_j |x_i - c_j|^2 (ties: smallest j)
__global__ __launch_bounds__(256) void k_assign(const double* __restrict__ x, int64_t n,
                                                const double* __restrict__ c, int64_t m, int d,
                                                const double* __restrict__ xx, const double* __restrict__ cc,
                                                int* __restrict__ label, double* __restrict__ dist2) {
  __shared__ double xs[DK][TM + PADT];
  __shared__ double ys[DK][TN + PADT];
  __shared__ double redv[TM][17];
  __shared__ int redi[TM][17];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * TM;
  double best[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
  int bidx[4] = {0, 0, 0, 0};
  double xr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { int64_t r = row0 + ty * 4 + i; xr[i] = (r < n) ? xx[r] : 0.0; }
  for (int64_t col0 = 0; col0 < m; col0 += TN) {
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    for (int k0 = 0; k0 < d; k0 += DK) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int idx = tid + 256 * q, r = idx >> 4, k = idx & 15;
        double vx = 0.0, vy = 0.0;
        if (k0 + k < d) {
          if (row0 + r < n) vx = x[(row0 + r) * (int64_t)d + k0 + k];
          if (col0 + r < m) vy = c[(col0 + r) * (int64_t)d + k0 + k];
        }
        xs[k][r] = vx; ys[k][r] = vy;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < DK; ++k) {
        double a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = xs[k][ty * 4 + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = ys[k][tx * 4 + j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t cj = col0 + tx * 4 + j;
      if (cj >= m) continue;
      const double cn = cc[cj];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double sq = xr[i] - 2.0 * acc[i][j] + cn;
        if (sq < best[i]) { best[i] = sq; bidx[i] = (int)cj; }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { redv[ty * 4 + i][tx] = best[i]; redi[ty * 4 + i][tx] = bidx[i]; }
  __syncthreads();
  if (tid < TM) {
    double s = INFINITY;
    int bi = 0;
    for (int t = 0; t < 16; ++t) {
      const double v = redv[tid][t];
      const int ix = redi[tid][t];
      if (v < s || (v == s && ix < bi)) { s = v; bi = ix; }
    }
    const int64_t r = row0 + tid;
    if (r < n) { label[r] = bi; dist2[r] = fmax(s, 0.0); }
  }
}

// The same assignment on the matrix cores (d <= 64), in the form of k_nn_distances_mfma (cov_kernels.hip):
// 8 waves x 16 cells per workgroup keep their MFMA A operands in registers; centre tiles of 64 are staged
// through LDS once per workgroup (double-buffered); running (min, argmin) per accumulator element.
typedef double v4d_k __attribute__((ext_vector_type(4)));
constexpr int KAS = 68;

__global__ __launch_bounds__(512) void k_assign_mfma(const double* __restrict__ x, int64_t n,
                                                     const double* __restrict__ c, int64_t m, int d,
                                                     const double* __restrict__ xx, const double* __restrict__ cc,
                                                     int* __restrict__ label, double* __restrict__ dist2) {
  __shared__ double ys[2][TN * KAS];
  __shared__ double yn[2][TN];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 16;
  const int ksteps = (d + 3) / 4;
  double a[16];
  {
    const int64_t ar = (row0 + li < n) ? row0 + li : n - 1;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int k = 4 * ks + lk;
      a[ks] = (ks < ksteps && k < d) ? x[ar * d + k] : 0.0;
    }
  }
  double xr[4], best[4];
  int bidx[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = row0 + lk + 4 * r;
    xr[r] = (row < n) ? xx[row] : 0.0;
    best[r] = INFINITY;
    bidx[r] = 0;
  }
  for (int e = tid; e < 2 * TN * KAS; e += 512) (&ys[0][0])[e] = 0.0;
  __syncthreads();
  auto stage = [&](int buf, int64_t col0) {
    const int cnt = TN * d;
    for (int e = tid; e < cnt; e += 512) {
      const int r = e / d, k = e - r * d;
      ys[buf][r * KAS + k] = (col0 + r < m) ? c[(col0 + r) * d + k] : 0.0;
    }
    if (tid < TN) yn[buf][tid] = (col0 + tid < m) ? cc[col0 + tid] : 0.0;
  };
  stage(0, 0);
  __syncthreads();
  int buf = 0;
  for (int64_t col0 = 0; col0 < m; col0 += TN, buf ^= 1) {
    if (col0 + TN < m) stage(buf ^ 1, col0 + TN);
    v4d_k acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = v4d_k{0.0, 0.0, 0.0, 0.0};
    const double* yb = &ys[buf][li * KAS + lk];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks < ksteps) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], yb[16 * t * KAS + 4 * ks], acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {   // ascending column order: strict < keeps the smallest index on ties
      const int64_t cj = col0 + 16 * t + li;
      const double cn = yn[buf][16 * t + li];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double sq = xr[r] - 2.0 * acc[t][r] + cn;
        if (cj < m && sq < best[r]) { best[r] = sq; bidx[r] = (int)cj; }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double s_ = best[r];
    int bi = bidx[r];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      const double so = __shfl_xor(s_, off, 64);
      const int io = __shfl_xor(bi, off, 64);
      if (so < s_ || (so == s_ && io < bi)) { s_ = so; bi = io; }
    }
    const int64_t row = row0 + lk + 4 * r;
    if (li == 0 && row < n) { label[row] = bi; dist2[row] = fmax(s_, 0.0); }
  }
}

// Cluster sums in FIXED POINT (round 4c).  The sums were fp64 atomics -- summation order not fixed, centres reproducible to
// rounding only, and with them every landmark set mln_kmeans returned (four calls, four checksums: the default call of the
// estimator on a large input was the same bits from run to run in everything but its landmarks).  Integer addition is
// associative: each coordinate goes in as llrint(x * 2^e_k), e_k per column such that n cells of the column's largest
// magnitude cannot overflow 62 bits, and comes out as sum * 2^-e_k / count.  Resolution: |x|_max,k * n * 2^-62 per
// coordinate -- 2e-13 of the column's range at 1e6 cells; a cell that leaves a cluster takes out exactly what it put in.
__global__ void k_km_colmax(const double* __restrict__ x, int64_t n, int d, unsigned long long* __restrict__ colmax) {
  // The first (blockDim / d) * d threads of every workgroup walk the row-major matrix with a stride that is a multiple of d:
  // a thread's column never changes, its maximum stays in a register, ONE atomic per thread at the end.  (Round 4 issued an
  // atomicMax per ELEMENT: 16 ms at 1e6 x 50, twice per k-means.)
  const int usable = ((int)blockDim.x / d) * d;
  if ((int)threadIdx.x >= usable) return;
  const int64_t total = n * d, T = (int64_t)gridDim.x * usable;
  double mx = 0.0;
  bool bad = false;
  for (int64_t e = (int64_t)blockIdx.x * usable + threadIdx.x; e < total; e += T) {
    const double v = fabs(x[e]);
    if (v < INFINITY) mx = fmax(mx, v); else bad = true;     // NaN / inf: the fixed-point sums (llrint) would turn it into garbage silently
  }
  const int col = (int)(((int64_t)blockIdx.x * usable + threadIdx.x) % d);
  if (mx > 0.0) atomicMax(&colmax[col], (unsigned long long)__double_as_longlong(mx));   // non-negative doubles order as integers
  if (bad) atomicMax(&colmax[d], 1ull);
}
__global__ void k_km_colscale(const unsigned long long* __restrict__ colmax, int d, int64_t n, double* __restrict__ scale) {
  for (int k = threadIdx.x; k < d; k += blockDim.x) {
    const double a = __longlong_as_double((long long)colmax[k]);
    int e = 0;
    if (a > 0.0 && a < INFINITY) {
      int ea = 0, en = 0;
      (void)frexp(a, &ea);                    // a < 2^ea
      (void)frexp((double)n, &en);            // n < 2^en
      e = 62 - ea - en;
      e = e > 1000 ? 1000 : (e < -1000 ? -1000 : e);
    }
    scale[k] = ldexp(1.0, e);
    scale[d + k] = ldexp(1.0, -e);
  }
}

__global__ void k_accumulate(const double* __restrict__ x, int64_t n, int d, const int* __restrict__ label,
                             double* __restrict__ sums, double* __restrict__ counts, const double* __restrict__ colscale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.y + threadIdx.y;
  if (i >= n) return;
  const int l = label[i];
  unsigned long long* isums = reinterpret_cast<unsigned long long*>(sums);
  for (int k = threadIdx.x; k < d; k += blockDim.x)
    atomicAdd(&isums[(int64_t)l * d + k], (unsigned long long)llrint(x[i * (int64_t)d + k] * colscale[k]));
  if (threadIdx.x == 0) atomicAdd(&counts[l], 1.0);      // (+-1 in fp64 is exact: any order, the same count)
}

// new centres (empty clusters keep their previous centre); sq[j] = |new - old|^2 (summed in fixed order by k_km_sum_fixed)
// delta (optional): how far each centre moved (0 for an empty cluster) -- what the distance bounds of the next sweep need
__global__ void k_finish(double* __restrict__ c, const double* __restrict__ sums, const double* __restrict__ counts,
                         int64_t m, int d, double* __restrict__ sq, double* __restrict__ delta, const double* __restrict__ colscale) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const double cnt = counts[j];
  if (cnt <= 0.0) { if (delta) delta[j] = 0.0; sq[j] = 0.0; return; }
  const long long* isums = reinterpret_cast<const long long*>(sums);
  double s = 0.0;
  for (int k = 0; k < d; ++k) {
    const double nv = (double)isums[j * d + k] * colscale[d + k] / cnt, t = nv - c[j * d + k];
    s = fma(t, t, s);
    c[j * d + k] = nv;
  }
  if (delta) delta[j] = sqrt(s);
  sq[j] = s;
}
// kstate (sweeps queued ahead of the host, see kmeans_level): [0] converged, [1] sweeps counted.  A counted sweep whose total
// movement is <= tol is the last: the kernels of the sweeps queued behind it find nothing to do.
__global__ __launch_bounds__(256) void k_km_sum_fixed(const double* __restrict__ v, int64_t m, double* __restrict__ out,
                                                      double tol = 0.0, int* __restrict__ kstate = nullptr) {
  __shared__ double part[256];
  double s = 0.0;
  for (int64_t j = threadIdx.x; j < m; j += 256) s += v[j];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = part[0];
    if (kstate && !kstate[0]) {
      kstate[1] += 1;
      if (part[0] <= tol) kstate[0] = 1;
    }
  }
}

// ---- Lloyd's iterations with distance bounds (Hamerly 2010: the same assignments, most of them without a search) --------
// Per cell: ub >= distance to the centre it is assigned to, lb <= distance to every OTHER centre.  When the centres move by
// delta_j, ub grows by delta of the own centre and lb shrinks by the largest movement among the others; while ub <= lb the
// assignment cannot have changed and the cell is skipped.  Otherwise ub is first tightened to the exact distance (d flops),
// and only if that does not decide either is the cell searched again (rowmin_f16.hip, restricted to the listed rows).
// dstat: [0] largest delta, [1] second largest, [2] index of the largest.
__global__ __launch_bounds__(256) void k_km_delta_stats(const double* __restrict__ delta, int64_t m, double* __restrict__ dstat) {
  __shared__ double s1[256], s2[256];
  __shared__ int sa[256];
  double b1 = 0.0, b2 = 0.0;
  int a = 0;
  for (int64_t j = threadIdx.x; j < m; j += 256) {
    const double v = delta[j];
    if (v > b1) { b2 = b1; b1 = v; a = (int)j; }
    else if (v > b2) b2 = v;
  }
  s1[threadIdx.x] = b1; s2[threadIdx.x] = b2; sa[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double g1 = 0.0, g2 = 0.0;
    int ga = 0;
    for (int t = 0; t < 256; ++t) {
      if (s1[t] > g1) { g2 = fmax(g1, s2[t]); g1 = s1[t]; ga = sa[t]; }
      else g2 = fmax(g2, s1[t]);
    }
    dstat[0] = g1; dstat[1] = g2; dstat[2] = (double)ga;
  }
}

// Eight lanes per cell (each reads every eighth coordinate: a wave touches 8 consecutive rows of x, coalesced); the exact
// distance to the own centre is taken for EVERY cell (n d reads: a tenth of a search sweep), so ub is always tight.
// A workgroup walks KB_ROWS cells and writes its flagged ones, in row order, to ITS segment of flagged_tmp and their number to
// wg_count; k_km_flag_scan / k_km_flag_compact close the gaps: the list of open rows is in row order (the same from run to
// run -- the pruned sweep's row blocks are cut from it), with no atomics.
constexpr int KB_ROWS = 512, KB_IT = KB_ROWS / 32;
__global__ __launch_bounds__(256) void k_km_bounds(const double* __restrict__ x, int64_t n, int d, const double* __restrict__ c,
                                                   const int* __restrict__ label, double* __restrict__ ub, double* __restrict__ lb,
                                                   const double* __restrict__ delta, const double* __restrict__ dstat,
                                                   const int* __restrict__ kstate, int* __restrict__ wg_count,
                                                   int* __restrict__ flagged_tmp) {
  __shared__ unsigned long long masks[KB_IT * 4];     // per (iteration, wave): ballot of the flagged groups' first lanes
  __shared__ int offs[KB_IT * 4];
  if (kstate[0]) {                                    // converged: no row is open
    if (threadIdx.x == 0) wg_count[blockIdx.x] = 0;
    return;
  }
  const int sub = threadIdx.x & 7, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double d1 = dstat[0], d2 = dstat[1];
  const int amax = (int)dstat[2];
  const int64_t row0 = (int64_t)blockIdx.x * KB_ROWS;
  for (int it = 0; it < KB_IT; ++it) {
    const int64_t i = row0 + it * 32 + (threadIdx.x >> 3);
    bool flag = false;
    if (i < n) {
      const int a = label[i];
      const double* xr = x + i * d;
      const double* cr = c + (int64_t)a * d;
      double dd = 0.0;
      for (int k = sub; k < d; k += 8) { const double t = xr[k] - cr[k]; dd = fma(t, t, dd); }
      dd += __shfl_xor(dd, 1, 64);
      dd += __shfl_xor(dd, 2, 64);
      dd += __shfl_xor(dd, 4, 64);
      const double u = sqrt(dd);
      const double l = lb[i] - ((a == amax) ? d2 : d1);
      if (sub == 0) { ub[i] = u; lb[i] = l; }
      flag = (sub == 0) && (u > l);
    }
    const unsigned long long mask = __ballot(flag);
    if (lane == 0) masks[it * 4 + wave] = mask;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int e = 0; e < KB_IT * 4; ++e) { offs[e] = run; run += __popcll(masks[e]); }
    wg_count[blockIdx.x] = run;
  }
  __syncthreads();
  if (threadIdx.x < KB_IT * 4) {
    const int e = threadIdx.x, it = e >> 2, w = e & 3;
    unsigned long long mk = masks[e];
    int slot = blockIdx.x * KB_ROWS + offs[e];
    while (mk) {
      const int ln = __ffsll((long long)mk) - 1;
      mk &= mk - 1;
      flagged_tmp[slot++] = (int)(row0 + it * 32 + w * 8 + (ln >> 3));
    }
  }
}

// exclusive prefix sums of the workgroups' counts (one workgroup; nwg <= 2^16 at 2^25 cells), total -> n_flag
__global__ __launch_bounds__(1024) void k_km_flag_scan(const int* __restrict__ wg_count, int nwg, int* __restrict__ wg_off,
                                                       int* __restrict__ n_flag) {
  __shared__ int part[1024];
  const int per = (nwg + 1023) / 1024, lo = threadIdx.x * per, hi = (lo + per < nwg) ? lo + per : nwg;
  int s = 0;
  for (int e = lo; e < hi; ++e) s += wg_count[e];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = 1; w < 1024; w <<= 1) {
    const int v = ((int)threadIdx.x >= w) ? part[threadIdx.x - w] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - s;
  for (int e = lo; e < hi; ++e) { wg_off[e] = run; run += wg_count[e]; }
  if (threadIdx.x == 1023) n_flag[0] = part[1023];
}
__global__ __launch_bounds__(256) void k_km_flag_compact(const int* __restrict__ flagged_tmp, const int* __restrict__ wg_count,
                                                         const int* __restrict__ wg_off, int* __restrict__ flagged) {
  const int cnt = wg_count[blockIdx.x], off = wg_off[blockIdx.x];
  for (int k = threadIdx.x; k < cnt; k += 256) flagged[off + k] = flagged_tmp[blockIdx.x * KB_ROWS + k];
}

// ---- group bounds (round 6): which 256-centre stages of the sweep can hold a closer centre ---------------------------------------
// (Ding et al. 2015, "Yinyang k-means", with the sweep's stages as the groups.)  The centres are swept in a fixed geometric
// order (cperm: recursive bisection of the first centres along the coordinate of largest spread, cut at multiples of 256 -- a
// stage of the sweep is a compact group of centres) and the cells are stored in the order of their first label's position, so
// a 256-row block of the open-row list has few distinct labels, all close.  Per cell i and stage s,
//     lbg[i][s] - cum[s]  <=  |x_i - c_j|   for every centre j of stage s other than the cell's label,
// cum[s] the sum over the sweeps so far of the largest movement of a centre of the stage: a bound written once stays valid
// without being touched (no n x S update per sweep).  A cell is open when its distance to its own centre exceeds one of
// its bounds, and needs exactly the stages of those bounds; a row block sweeps the union.  Two sweeps over the open rows:
// the winner sweep (label, second smallest value) and the stage-minimum sweep (rowmin_w64.hip, SMIN), from which
// k_km_resolve renews the bounds of the swept stages.  On 1e6 cells / 5000 centres the Hamerly bound above leaves ~30 % of
// the cells open, each against all 20 stages.
__global__ void k_km_gather_centres(const double* __restrict__ c, const int* __restrict__ cperm, int64_t m, int d,
                                    double* __restrict__ cp) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m * d) return;
  const int64_t p = t / d;
  cp[t] = c[(int64_t)cperm[p] * d + (t - p * d)];
}
// one workgroup per stage: cum[s] += the largest movement among the stage's centres
__global__ __launch_bounds__(256) void k_km_stage_delta(const double* __restrict__ delta, const int* __restrict__ cperm, int64_t m,
                                                        double* __restrict__ cum) {
  __shared__ double red[256];
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  red[threadIdx.x] = (p < m) ? delta[cperm[p]] : 0.0;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + w]);
    __syncthreads();
  }
  if (threadIdx.x == 0) cum[blockIdx.x] += red[0];
}
// k_km_bounds with group bounds: the exact distance to the own centre against every stage's bound; rowmask[i] = the stages
// the cell needs (written for open cells).  S <= 32.
__global__ __launch_bounds__(256) void k_km_bounds_g(const double* __restrict__ x, int64_t n, int d, const double* __restrict__ c,
                                                     const int* __restrict__ label, double* __restrict__ ub,
                                                     const float* __restrict__ lbg, const double* __restrict__ cum, int S,
                                                     const int* __restrict__ cpos, const int* __restrict__ kstate,
                                                     int* __restrict__ wg_count, int* __restrict__ flagged_tmp,
                                                     uint32_t* __restrict__ rowmask) {
  __shared__ unsigned long long masks[KB_IT * 4];
  __shared__ int offs[KB_IT * 4];
  if (kstate[0]) {
    if (threadIdx.x == 0) wg_count[blockIdx.x] = 0;
    return;
  }
  const int sub = threadIdx.x & 7, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t row0 = (int64_t)blockIdx.x * KB_ROWS;
  for (int it = 0; it < KB_IT; ++it) {
    const int64_t i = row0 + it * 32 + (threadIdx.x >> 3);
    bool flag = false;
    if (i < n) {
      const int a = label[i];
      const double* xr = x + i * d;
      const double* cr = c + (int64_t)a * d;
      double dd = 0.0;
      for (int k = sub; k < d; k += 8) { const double t = xr[k] - cr[k]; dd = fma(t, t, dd); }
      dd += __shfl_xor(dd, 1, 64);
      dd += __shfl_xor(dd, 2, 64);
      dd += __shfl_xor(dd, 4, 64);
      const double u = sqrt(dd);
      uint32_t bits = 0u;
      for (int s = sub; s < S; s += 8) bits |= ((double)lbg[i * S + s] - cum[s] < u) ? (1u << s) : 0u;
      bits |= __shfl_xor(bits, 1, 64);
      bits |= __shfl_xor(bits, 2, 64);
      bits |= __shfl_xor(bits, 4, 64);
      // (an open cell's own stage is always swept: the new label is the sweep's winner, and the present label competes)
      if (sub == 0) { ub[i] = u; if (bits) rowmask[i] = bits | (1u << (cpos[a] >> 8)); }
      flag = (sub == 0) && bits != 0u;
    }
    const unsigned long long mask = __ballot(flag);
    if (lane == 0) masks[it * 4 + wave] = mask;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int e = 0; e < KB_IT * 4; ++e) { offs[e] = run; run += __popcll(masks[e]); }
    wg_count[blockIdx.x] = run;
  }
  __syncthreads();
  if (threadIdx.x < KB_IT * 4) {
    const int e = threadIdx.x, it = e >> 2, w = e & 3;
    unsigned long long mk = masks[e];
    int slot = blockIdx.x * KB_ROWS + offs[e];
    while (mk) {
      const int ln = __ffsll((long long)mk) - 1;
      mk &= mk - 1;
      flagged_tmp[slot++] = (int)(row0 + it * 32 + w * 8 + (ln >> 3));
    }
  }
}
// one workgroup per 256 rows of the open-row list: the union of the stages its rows need
__global__ __launch_bounds__(256) void k_km_block_mask_g(const int* __restrict__ flagged, const int* __restrict__ n_flag,
                                                         const uint32_t* __restrict__ rowmask, uint32_t* __restrict__ mask) {
  __shared__ uint32_t acc;
  const int F = n_flag[0];
  if ((int64_t)blockIdx.x * 256 >= F) return;
  if (threadIdx.x == 0) acc = 0u;
  __syncthreads();
  const int r = blockIdx.x * 256 + threadIdx.x;
  uint32_t bits = (r < F) ? rowmask[flagged[r]] : 0u;
  for (int off = 1; off < 64; off <<= 1) bits |= __shfl_xor(bits, off, 64);
  if ((threadIdx.x & 63) == 0) atomicOr(&acc, bits);
  __syncthreads();
  if (threadIdx.x == 0) mask[blockIdx.x] = acc;
}
__global__ void k_km_gather_rows(const double* __restrict__ x, const int* __restrict__ perm, int64_t n, int d, double* __restrict__ xs) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * d) return;
  const int64_t r = t / d;
  xs[t] = x[(int64_t)perm[r] * d + (t - r * d)];
}
__global__ void k_km_gather_state(const int* __restrict__ perm, int64_t n, const int* __restrict__ label, const double* __restrict__ ub,
                                  const double* __restrict__ lb, int* __restrict__ label_s, double* __restrict__ ub_s,
                                  double* __restrict__ lb_s) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int i = perm[r];
  label_s[r] = label[i]; ub_s[r] = ub[i]; lb_s[r] = lb[i];
}

// variance of column blockIdx.x over the rows 0, stride, 2 stride, ... (ns of them): two passes, sums in a fixed order
__global__ __launch_bounds__(1024) void k_km_col_var(const double* __restrict__ x, int64_t ns, int64_t stride, int d, double* __restrict__ var) {
  __shared__ double part[1024];
  const int k = blockIdx.x;
  auto block_sum = [&](double v) {
    part[threadIdx.x] = v;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
      __syncthreads();
    }
    const double r = part[0];
    __syncthreads();
    return r;
  };
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < ns; i += 1024) s += x[i * stride * d + k];
  const double mean = block_sum(s) / (double)ns;
  double m2 = 0.0;
  for (int64_t i = threadIdx.x; i < ns; i += 1024) { const double t = x[i * stride * d + k] - mean; m2 = fma(t, t, m2); }
  const double tot = block_sum(m2);
  if (threadIdx.x == 0) var[k] = tot / (double)ns;
}

struct XorShift {
  unsigned long long s;
  explicit XorShift(unsigned long long seed) : s(seed * 0x9E3779B97F4A7C15ULL + 0xD1B54A32D192ED03ULL) { next(); next(); }
  unsigned long long next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
  double uniform() { return (double)(next() >> 11) / 9007199254740992.0; }
};
}  // namespace

// The order in which the pruned sweeps walk the centres: recursive bisection along the coordinate of largest variance, cut at
// multiples of 256 (a stage of the sweep) and, inside a stage, of 32 (a leaf); ids ascending inside a leaf.
static void centre_sweep_order(const double* c, int m, int d, int* order) {
  for (int j = 0; j < m; ++j) order[j] = j;
  struct Seg { int lo, hi; };
  std::vector<Seg> todo{{0, m}};
  while (!todo.empty()) {
    const Seg sg = todo.back();
    todo.pop_back();
    const int len = sg.hi - sg.lo;
    if (len <= 32) { std::sort(order + sg.lo, order + sg.hi); continue; }
    const int unit = len > 256 ? 256 : 32;
    int best = 0;
    double best_var = -1.0;
    for (int k = 0; k < d; ++k) {
      double mean = 0.0, var = 0.0;
      for (int p = sg.lo; p < sg.hi; ++p) mean += c[(size_t)order[p] * d + k];
      mean /= len;
      for (int p = sg.lo; p < sg.hi; ++p) { const double t = c[(size_t)order[p] * d + k] - mean; var += t * t; }
      if (var > best_var) { best_var = var; best = k; }
    }
    const int left = (((len + unit - 1) / unit) / 2) * unit;
    std::nth_element(order + sg.lo, order + sg.lo + left, order + sg.hi, [&](int a, int b) {
      const double va = c[(size_t)a * d + best], vb = c[(size_t)b * d + best];
      return va < vb || (va == vb && a < b);
    });
    todo.push_back({sg.lo, sg.lo + left});
    todo.push_back({sg.lo + left, sg.hi});
  }
}

// One level: k-means++ seeding (or the centres in `init`, device, m x d) followed by Lloyd's iterations on the n cells at x.
static int kmeans_level(mln_ctx* ctx, const double* x, int64_t n, int32_t d, int64_t m, int64_t seed,
                        int32_t max_iter, double tol, const double* init, double* centers, int32_t* n_iter_out,
                        double* inertia_out) {
  hipStream_t st = ctx->stream;
  double *dx = nullptr, *dc = nullptr, *xx = nullptr, *cc = nullptr, *mind = nullptr, *bsum = nullptr, *sums = nullptr,
         *counts = nullptr, *shift = nullptr;
  int* label = nullptr;
  int64_t* pick = nullptr;
  bool own_x = false;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, x) == hipSuccess && (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged)) {
    dx = const_cast<double*>(x);
  } else {
    (void)hipGetLastError();
    MLN_HIP(ctx, mln_dmalloc((void**)&dx, sizeof(double) * (size_t)n * d));
    own_x = true;
    MLN_HIP(ctx, hipMemcpyAsync(dx, x, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice, st));
  }
  const int64_t nblk = (n + SBLK - 1) / SBLK, nblk_h = (n + 255) / 256;     // (block sums of the fp64 / the half-precision seeding)
  MLN_HIP(ctx, mln_dmalloc((void**)&dc, sizeof(double) * (size_t)m * d));
  MLN_HIP(ctx, mln_dmalloc((void**)&xx, sizeof(double) * (size_t)n));
  MLN_HIP(ctx, mln_dmalloc((void**)&cc, sizeof(double) * (size_t)m));
  MLN_HIP(ctx, mln_dmalloc((void**)&mind, sizeof(double) * (size_t)n));
  MLN_HIP(ctx, mln_dmalloc((void**)&bsum, sizeof(double) * (size_t)nblk_h));
  MLN_HIP(ctx, mln_dmalloc((void**)&sums, sizeof(double) * (size_t)m * d));
  MLN_HIP(ctx, mln_dmalloc((void**)&counts, sizeof(double) * (size_t)m));
  MLN_HIP(ctx, mln_dmalloc((void**)&shift, sizeof(double)));
  // fixed-point cluster sums: per-column magnitudes -> power-of-two scales [0, d) and their inverses [d, 2d); sq: |shift|^2 per centre
  unsigned long long* colmax = nullptr;
  double *colscale = nullptr, *sq = nullptr;
  MLN_HIP(ctx, mln_dmalloc((void**)&colmax, sizeof(unsigned long long) * (size_t)(d + 1)));
  MLN_HIP(ctx, mln_dmalloc((void**)&colscale, sizeof(double) * 2 * (size_t)d));
  MLN_HIP(ctx, mln_dmalloc((void**)&sq, sizeof(double) * (size_t)(m > d ? m : d)));     // (also the d column variances)
  MLN_HIP(ctx, mln_dmalloc((void**)&label, sizeof(int) * (size_t)n));
  MLN_HIP(ctx, mln_dmalloc((void**)&pick, sizeof(int64_t)));
  int rc = MLN_OK;
  auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == MLN_OK) rc = mln_hip_fail(ctx, e, "kmeans", __FILE__, __LINE__); };

  chk(hipMemsetAsync(colmax, 0, sizeof(unsigned long long) * (size_t)(d + 1), st));
  if (rc == MLN_OK) {
    hipLaunchKernelGGL(k_km_colmax, dim3(1024), dim3(256), 0, st, dx, n, d, colmax);
    hipLaunchKernelGGL(k_km_colscale, dim3(1), dim3(256), 0, st, colmax, d, n, colscale);
    unsigned long long bad = 0;
    chk(hipMemcpyAsync(&bad, colmax + d, sizeof(bad), hipMemcpyDeviceToHost, st));
    chk(hipStreamSynchronize(st));
    if (rc == MLN_OK && bad) { mln_set_error(ctx, "kmeans: x holds non-finite values"); rc = MLN_ERR_ARG; }
  }
  // ---- k-means++ seeding ------------------------------------------------------------------------
  // m - 1 sequential draws, each after one update of every cell's distance to its nearest centre so far.  Everything stays
  // on the device (the uniform draws are uploaded once); large problems read the half-precision copy of the cells.
  XorShift rng((unsigned long long)seed);
  int64_t cur = (int64_t)(rng.uniform() * (double)n);
  if (cur >= n) cur = n - 1;
  const bool seeded = init == nullptr;
  const bool km_fp16_seed = !(mln_experiment("MELLON_AMD_KM_FP16") && std::atoi(mln_experiment("MELLON_AMD_KM_FP16")) == 0);
  const bool seed_h = km_fp16_seed && d <= 61 && n * m >= ((int64_t)1 << 24) && m >= 2;
  void* xsplit = nullptr;
  double* prep = nullptr;          // centre and scale of the half-precision copies (rowmin_prepare)
  if (d <= 64 && n * m >= ((int64_t)1 << 24) && m >= 2 && rc == MLN_OK) {
    chk(mln_dmalloc((void**)&prep, sizeof(double) * ROWMIN_PREP_DOUBLES));
    if (rc == MLN_OK) rc = rowmin_prepare(ctx, dx, n, nullptr, 0, d, prep);
  }
  double* xxs = nullptr;           // squared norms of the centred, scaled cells (the error bound of the fp16 sweep needs them)
  if (prep && rc == MLN_OK) chk(mln_dmalloc((void**)&xxs, sizeof(double) * (size_t)n));
  if (seed_h && rc == MLN_OK) {
    chk(mln_dmalloc(&xsplit, rowmin_split_bytes(n)));
    if (rc == MLN_OK) rc = launch_split_f16(ctx, dx, n, d, xsplit, xxs, nullptr, 1, prep);   // role 1: -2 x (also the Lloyd sweeps' operand)
  }
  if (rc == MLN_OK && seeded) chk(hipMemcpyAsync(dc, dx + cur * d, sizeof(double) * d, hipMemcpyDeviceToDevice, st));
  if (rc == MLN_OK && !seeded) chk(hipMemcpyAsync(dc, init, sizeof(double) * (size_t)m * d, hipMemcpyDeviceToDevice, st));
  for (int64_t j = 0; seeded && j + 1 < m && rc == MLN_OK; ++j) {
    if (seed_h) {
      hipLaunchKernelGGL((k_seed_update_h<256>), dim3((unsigned)nblk_h), dim3(1024), 0, st, reinterpret_cast<const _Float16*>(xsplit), n, d,
                         -0.5f, dc + j * d, prep, mind, bsum, j == 0 ? 1 : 0);
      hipLaunchKernelGGL((k_seed_select<256>), dim3(1), dim3(256), 0, st, bsum, nblk_h, mind, n, rng.uniform(), dx, d, dc + (j + 1) * d);
    } else {
      hipLaunchKernelGGL(k_seed_update, dim3((unsigned)nblk), dim3(256), 0, st, dx, n, d, dc + j * d, mind, bsum, j == 0 ? 1 : 0);
      hipLaunchKernelGGL((k_seed_select<SBLK>), dim3(1), dim3(256), 0, st, bsum, nblk, mind, n, rng.uniform(), dx, d, dc + (j + 1) * d);
    }
    if ((j & 1023) == 1023) chk(hipStreamSynchronize(st));     // (bounds the launch queue)
  }
  chk(hipGetLastError());

  // ---- Lloyd ------------------------------------------------------------------------------------------
  // tolerance scaled like sklearn: tol * mean over features of the feature variance
  double scaled_tol = 0.0;
  if (rc == MLN_OK) {
    const int64_t ns = (n < 200000) ? n : 200000;   // variance estimate from an evenly spaced subset (on the device: the
    const int64_t stride = n / ns;                   // 32 MB copy to the host and its loops were 8-14 ms per level)
    std::vector<double> hv((size_t)d);
    hipLaunchKernelGGL(k_km_col_var, dim3((unsigned)d), dim3(1024), 0, st, dx, ns, stride, d, sq);     // (sq: max(m, d) doubles)
    chk(hipMemcpyAsync(hv.data(), sq, sizeof(double) * (size_t)d, hipMemcpyDeviceToHost, st));
    chk(hipStreamSynchronize(st));
    double var_sum = 0.0;
    for (int k = 0; k < d; ++k) var_sum += hv[(size_t)k];
    scaled_tol = tol * var_sum / d;
  }
  int it = 0;
  hipLaunchKernelGGL(k_sqnorm_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dx, n, d, xx);
  // Assignment on the fp16 matrix cores (rowmin_f16.hip: cells and centres split into hi + lo halves, three products,
  // fp32 accumulation: distances to ~1e-5 relative).  A cell whose two nearest centres tie within that may land on
  // either -- k-means is indifferent -- and the centre update, the stopping test and the final inertia stay fp64.
  // 5e11 fp64 flops per sweep at 1e6 cells x 5000 centres (15 ms) become ~1 ms.  MELLON_AMD_KM_FP16=0 disables.
  const bool km_fp16 = !(mln_experiment("MELLON_AMD_KM_FP16") && std::atoi(mln_experiment("MELLON_AMD_KM_FP16")) == 0);
  const bool fast_assign = km_fp16 && d <= 64 && n * m >= ((int64_t)1 << 24) && m >= 2;
  const bool km_fold = d <= 61;
  void* csplit = nullptr;
  float *ccf = nullptr, *m1f = nullptr;
  if (fast_assign && rc == MLN_OK) {
    const bool have = xsplit != nullptr;       // (the seeding's copy has role 1: right for the folded product)
    if (!have) chk(mln_dmalloc(&xsplit, rowmin_split_bytes(n)));
    chk(mln_dmalloc(&csplit, rowmin_split_bytes(m)));
    chk(mln_dmalloc((void**)&ccf, sizeof(float) * (size_t)m));
    chk(mln_dmalloc((void**)&m1f, sizeof(float) * (size_t)n));
    if (rc == MLN_OK && !have) rc = launch_split_f16(ctx, dx, n, d, xsplit, xxs, nullptr, km_fold ? 1 : 0, prep);
  }
  // With the folded fp16 sweep available the iterations carry distance bounds (see k_km_bounds): sweep 0 searches every
  // cell, later sweeps only the cells whose bounds no longer decide; the sums of the clusters follow the cells that moved.
  const bool km_bounds = fast_assign && km_fold && xxs && !(mln_experiment("MELLON_AMD_KM_BOUNDS") && std::atoi(mln_experiment("MELLON_AMD_KM_BOUNDS")) == 0);
  double *ub = nullptr, *lb = nullptr, *delta = nullptr, *dstat = nullptr, *yy = nullptr, *ymax = nullptr;
  float* m2f = nullptr;
  int *argc = nullptr, *nflag = nullptr, *flagged = nullptr;
  if (km_bounds && rc == MLN_OK) {
    chk(mln_dmalloc((void**)&ub, sizeof(double) * (size_t)n));
    chk(mln_dmalloc((void**)&lb, sizeof(double) * (size_t)n));
    chk(mln_dmalloc((void**)&delta, sizeof(double) * (size_t)m));
    chk(mln_dmalloc((void**)&dstat, sizeof(double) * 4));
    chk(mln_dmalloc((void**)&yy, sizeof(double) * (size_t)m));
    chk(mln_dmalloc((void**)&ymax, sizeof(double)));
    chk(mln_dmalloc((void**)&m2f, sizeof(float) * (size_t)n));
    chk(mln_dmalloc((void**)&argc, sizeof(int) * (size_t)n));
    chk(mln_dmalloc((void**)&nflag, sizeof(int)));
    chk(mln_dmalloc((void**)&flagged, sizeof(int) * (size_t)n));
  }
  // Group bounds (see k_km_gather_centres): 1024 to 8192 centres (at most 32 stages).  MELLON_AMD_KM_PRUNE=0 disables.
  const int S = (int)((m + 255) / 256);
  const bool km_prune = km_bounds && m >= 1024 && S <= 32 && n >= 65536 && n <= 2147483647LL &&
                        !(mln_experiment("MELLON_AMD_KM_PRUNE") && std::atoi(mln_experiment("MELLON_AMD_KM_PRUNE")) == 0);
  int *kstate = nullptr, *wg_count = nullptr, *wg_off = nullptr, *flagged_tmp = nullptr, *cperm = nullptr, *cposd = nullptr, *perm = nullptr,
      *label_s = nullptr;
  double *dcp = nullptr, *cum = nullptr, *dxs = nullptr, *ub_s = nullptr, *lb_s = nullptr;
  float *lbg = nullptr, *smin = nullptr;
  uint32_t *smask = nullptr, *rowmask = nullptr;
  const int nwg = (int)((n + KB_ROWS - 1) / KB_ROWS);
  if (km_bounds && rc == MLN_OK) {
    chk(mln_dmalloc((void**)&kstate, sizeof(int) * 2));
    chk(mln_dmalloc((void**)&wg_count, sizeof(int) * (size_t)nwg));
    chk(mln_dmalloc((void**)&wg_off, sizeof(int) * (size_t)nwg));
    chk(mln_dmalloc((void**)&flagged_tmp, sizeof(int) * (size_t)nwg * KB_ROWS));
    if (rc == MLN_OK) chk(hipMemsetAsync(kstate, 0, sizeof(int) * 2, st));
  }
  if (km_prune && rc == MLN_OK) {
    chk(mln_dmalloc((void**)&cperm, sizeof(int) * (size_t)m));
    chk(mln_dmalloc((void**)&cposd, sizeof(int) * (size_t)m));
    chk(mln_dmalloc((void**)&perm, sizeof(int) * (size_t)n));
    chk(mln_dmalloc((void**)&label_s, sizeof(int) * (size_t)n));
    chk(mln_dmalloc((void**)&ub_s, sizeof(double) * (size_t)n));
    chk(mln_dmalloc((void**)&lb_s, sizeof(double) * (size_t)n));
    chk(mln_dmalloc((void**)&dcp, sizeof(double) * (size_t)m * d));
    chk(mln_dmalloc((void**)&cum, sizeof(double) * (size_t)S));
    chk(mln_dmalloc((void**)&dxs, sizeof(double) * (size_t)n * d));
    chk(mln_dmalloc((void**)&lbg, sizeof(float) * (size_t)n * S));
    chk(mln_dmalloc((void**)&smin, sizeof(float) * (size_t)n * S));
    chk(mln_dmalloc((void**)&smask, sizeof(uint32_t) * (size_t)((n + 255) / 256)));
    chk(mln_dmalloc((void**)&rowmask, sizeof(uint32_t) * (size_t)n));
    if (rc == MLN_OK) chk(hipMemsetAsync(cum, 0, sizeof(double) * (size_t)S, st));
  }
  if (km_bounds && rc == MLN_OK && max_iter > 0) {
    const double* xl = dx;        // the cells in the order the sweeps walk them (sorted by first label with group bounds)
    bool groups = false;
    KmGroups grp{lbg, cum, smin, n, cposd, S};
    // the part of a sweep after the assignment: new centres, their movement, the bounds, the list of open rows
    auto tail = [&]() {
      chk(hipMemsetAsync(shift, 0, sizeof(double), st));
      hipLaunchKernelGGL(k_finish, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, dc, sums, counts, m, d, sq, delta, colscale);
      hipLaunchKernelGGL(k_km_sum_fixed, dim3(1), dim3(256), 0, st, sq, m, shift, scaled_tol, kstate);
      if (groups) {
        hipLaunchKernelGGL(k_km_stage_delta, dim3((unsigned)S), dim3(256), 0, st, delta, cperm, m, cum);
        hipLaunchKernelGGL(k_km_bounds_g, dim3((unsigned)nwg), dim3(256), 0, st, xl, n, d, dc, label, ub, lbg, cum, S, cposd, kstate, wg_count,
                           flagged_tmp, rowmask);
      } else {
        hipLaunchKernelGGL(k_km_delta_stats, dim3(1), dim3(256), 0, st, delta, m, dstat);
        hipLaunchKernelGGL(k_km_bounds, dim3((unsigned)nwg), dim3(256), 0, st, xl, n, d, dc, label, ub, lb, delta, dstat, kstate, wg_count,
                           flagged_tmp);
      }
      hipLaunchKernelGGL(k_km_flag_scan, dim3(1), dim3(1024), 0, st, wg_count, nwg, wg_off, nflag);
      hipLaunchKernelGGL(k_km_flag_compact, dim3((unsigned)nwg), dim3(256), 0, st, flagged_tmp, wg_count, wg_off, flagged);
    };
    // ---- sweep 0: every cell against every centre (with group bounds: the centres already in their sweep order)
    if (km_prune) {
      std::vector<double> hc((size_t)m * d);
      std::vector<int> order((size_t)m), cpos((size_t)m);
      chk(hipMemcpyAsync(hc.data(), dc, sizeof(double) * (size_t)m * d, hipMemcpyDeviceToHost, st));
      chk(hipStreamSynchronize(st));
      if (rc == MLN_OK) {
        centre_sweep_order(hc.data(), (int)m, d, order.data());
        for (int p = 0; p < (int)m; ++p) cpos[(size_t)order[p]] = p;
        chk(hipMemcpyAsync(cperm, order.data(), sizeof(int) * (size_t)m, hipMemcpyHostToDevice, st));
        chk(hipMemcpyAsync(cposd, cpos.data(), sizeof(int) * (size_t)m, hipMemcpyHostToDevice, st));
        chk(hipStreamSynchronize(st));
        groups = rc == MLN_OK;
      }
    }
    auto centres_split = [&]() {
      if (groups) hipLaunchKernelGGL(k_km_gather_centres, dim3((unsigned)((m * d + 255) / 256)), dim3(256), 0, st, dc, cperm, m, d, dcp);
      int r = launch_split_f16(ctx, groups ? dcp : dc, m, d, csplit, yy, ccf, 2, prep);
      if (r == MLN_OK) r = launch_max_norm(ctx, yy, m, ymax);
      return r;
    };
    if (rc == MLN_OK) rc = centres_split();
    if (rc == MLN_OK) rc = launch_rowmin_masked(ctx, xsplit, n, nullptr, csplit, m, m1f, m2f, argc, nullptr, nullptr, 0);
    if (rc == MLN_OK) rc = launch_km_resolve(ctx, dx, n, nullptr, dc, m, d, xxs, ymax, prep, m2f, argc, label, ub, lb, nullptr, nullptr, colscale,
                                             nullptr, groups ? cperm : nullptr);
    if (groups && rc == MLN_OK) {
      // the order of the cells (a stable counting sort of the n first labels by sweep position: the same order from run to
      // run), once per call; then the stage minima of every cell and the first group bounds
      std::vector<int> hl((size_t)n), hpos((size_t)m), hperm((size_t)n);
      chk(hipMemcpyAsync(hl.data(), label, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st));
      chk(hipMemcpyAsync(hpos.data(), cposd, sizeof(int) * (size_t)m, hipMemcpyDeviceToHost, st));
      chk(hipStreamSynchronize(st));
      if (rc == MLN_OK) {
        std::vector<int64_t> start((size_t)m + 1, 0);
        for (int64_t i = 0; i < n; ++i) start[(size_t)hpos[(size_t)hl[(size_t)i]] + 1]++;
        for (int64_t p = 0; p < m; ++p) start[(size_t)p + 1] += start[(size_t)p];
        for (int64_t i = 0; i < n; ++i) hperm[(size_t)start[(size_t)hpos[(size_t)hl[(size_t)i]]]++] = (int)i;
        chk(hipMemcpyAsync(perm, hperm.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_km_gather_rows, dim3((unsigned)((n * d + 255) / 256)), dim3(256), 0, st, dx, perm, n, d, dxs);
        hipLaunchKernelGGL(k_km_gather_state, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, perm, n, label, ub, lb, label_s, ub_s, lb_s);
        chk(hipStreamSynchronize(st));                      // (the host vector above is the copy's source)
        if (rc == MLN_OK) rc = launch_split_f16(ctx, dxs, n, d, xsplit, xxs, nullptr, 1, prep);      // the same rows, in the new order
        std::swap(label, label_s); std::swap(ub, ub_s); std::swap(lb, lb_s);
        xl = dxs;
        if (rc == MLN_OK) rc = launch_rowmin_masked(ctx, xsplit, n, nullptr, csplit, m, nullptr, nullptr, nullptr, nullptr, nullptr, 0, smin, n);
        if (rc == MLN_OK) rc = launch_km_init_groups(ctx, n, label, lb, xxs, ymax, prep, &grp);
      }
    }
    if (rc == MLN_OK) {
      chk(hipMemsetAsync(sums, 0, sizeof(double) * (size_t)m * d, st));
      chk(hipMemsetAsync(counts, 0, sizeof(double) * (size_t)m, st));
      hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((n + 3) / 4)), dim3(64, 4), 0, st, xl, n, d, label, sums, counts, colscale);
      tail();
    }
    // ---- the sweeps over the open rows, queued eight at a time: their number (nflag) and the stopping test (kstate) stay
    // on the device -- a sweep behind the converged one finds no open row and moves nothing
    while (rc == MLN_OK) {
      int hk[2] = {0, 0};
      chk(hipMemcpyAsync(hk, kstate, sizeof(int) * 2, hipMemcpyDeviceToHost, st));
      chk(hipStreamSynchronize(st));
      it = hk[1];
      if (rc != MLN_OK || hk[0] || it >= max_iter) break;
      const int batch = (max_iter - it < 8) ? max_iter - it : 8;
      for (int b = 0; b < batch && rc == MLN_OK; ++b) {
        rc = centres_split();
        if (groups)
          hipLaunchKernelGGL(k_km_block_mask_g, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, flagged, nflag, rowmask, smask);
        if (rc == MLN_OK) rc = launch_rowmin_masked(ctx, xsplit, n, nflag, csplit, m, m1f, m2f, argc, flagged, groups ? smask : nullptr,
                                                    groups ? 1 : 0, groups ? smin : nullptr, n);       // (winner and stage minima in one sweep)
        if (rc == MLN_OK) rc = launch_km_resolve(ctx, xl, n, flagged, dc, m, d, xxs, ymax, prep, m2f, argc, label, ub, lb, sums, counts,
                                                 colscale, nflag, groups ? cperm : nullptr, groups ? &grp : nullptr, groups ? smask : nullptr, 1);
        if (rc == MLN_OK) tail();
      }
    }
  } else
  for (; it < max_iter && rc == MLN_OK; ++it) {
    hipLaunchKernelGGL(k_sqnorm_rows, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, dc, m, d, cc);
    if (fast_assign) {
      rc = launch_split_f16(ctx, dc, m, d, csplit, nullptr, ccf, km_fold ? 2 : 0, prep);
      if (rc == MLN_OK) rc = launch_rowmin_f16x3(ctx, xsplit, n, csplit, m, ccf, 0, 0, m1f, nullptr, label, km_fold ? 1 : 0);
      if (rc != MLN_OK) break;
    } else if (d <= 64 && n * m >= 4096)
      hipLaunchKernelGGL(k_assign_mfma, dim3((unsigned)((n + 127) / 128)), dim3(512), 0, st, dx, n, dc, m, d, xx, cc, label, mind);
    else
      hipLaunchKernelGGL(k_assign, dim3((unsigned)((n + TM - 1) / TM)), dim3(256), 0, st, dx, n, dc, m, d, xx, cc, label, mind);
    chk(hipMemsetAsync(sums, 0, sizeof(double) * (size_t)m * d, st));
    chk(hipMemsetAsync(counts, 0, sizeof(double) * (size_t)m, st));
    chk(hipMemsetAsync(shift, 0, sizeof(double), st));
    hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((n + 3) / 4)), dim3(64, 4), 0, st, dx, n, d, label, sums, counts, colscale);
    hipLaunchKernelGGL(k_finish, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, dc, sums, counts, m, d, sq, (double*)nullptr, colscale);
    hipLaunchKernelGGL(k_km_sum_fixed, dim3(1), dim3(256), 0, st, sq, m, shift, 0.0, (int*)nullptr);
    double hs = 0.0;
    chk(hipMemcpyAsync(&hs, shift, sizeof(double), hipMemcpyDeviceToHost, st));
    chk(hipStreamSynchronize(st));
    if (hs <= scaled_tol) { ++it; break; }
  }
  if (rc == MLN_OK && inertia_out) {   // sum of squared distances to the closest final centre
    hipLaunchKernelGGL(k_sqnorm_rows, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, dc, m, d, cc);
    if (d <= 64 && n * m >= 4096)
      hipLaunchKernelGGL(k_assign_mfma, dim3((unsigned)((n + 127) / 128)), dim3(512), 0, st, dx, n, dc, m, d, xx, cc, label, mind);
    else
      hipLaunchKernelGGL(k_assign, dim3((unsigned)((n + TM - 1) / TM)), dim3(256), 0, st, dx, n, dc, m, d, xx, cc, label, mind);
    std::vector<double> hm((size_t)n);
    chk(hipMemcpyAsync(hm.data(), mind, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
    chk(hipStreamSynchronize(st));
    double s = 0.0;
    for (double v : hm) s += v;
    *inertia_out = s;
  }
  if (rc == MLN_OK) {
    chk(hipGetLastError());
    chk(hipMemcpyAsync(centers, dc, sizeof(double) * (size_t)m * d, hipMemcpyDefault, st));
    chk(hipStreamSynchronize(st));
  }
  if (n_iter_out) *n_iter_out = it;
  (void)hipStreamSynchronize(st);
  void* ptrs[] = {dc, xx, cc, mind, bsum, sums, counts, shift, label, pick, xsplit, csplit, ccf, m1f, prep, xxs, ub, lb, delta, dstat, yy, ymax,
                  m2f, argc, nflag, flagged, colmax, colscale, sq, kstate, wg_count, wg_off, flagged_tmp, cperm, cposd, perm, label_s, ub_s,
                  lb_s, dcp, cum, dxs, lbg, smin, smask, rowmask};
  for (void* p : ptrs) if (p) (void)mln_dfree_synced(p);      // everything ran on st, synchronised above
  if (own_x) (void)mln_dfree_synced(dx);
  return rc;
}

// ---- sklearn-compatible k-means++ seeding (round 5) ---------------------------------------------------------------------
// What the reference's landmarks come from: sklearn.cluster.k_means(x, m, n_init=1, random_state=42) (parameters.py:275-291),
// i.e. _kmeans_plusplus with 2 + int(log m) local trials per centre, its random numbers from numpy's RandomState(seed):
//   first centre     rs.choice(n, p = uniform)
//   centre c         rand_vals = rs.uniform(size = L) * current_pot ; candidate_l = searchsorted(cumsum(closest_dist_sq), rand_vals)
//                    (clipped to n - 1) ; the candidate with the smallest potential sum_i min(closest_dist_sq_i, |x_i - cand|^2)
//                    becomes the centre, closest_dist_sq is updated with it.
// The binding draws the random numbers with numpy itself (the generator IS the specification) and hands them over; the device
// does the O(n L d) distances, the cumulative sums and the argmin -- two launches per centre:
//   k_sk_candidates   every cell: min(closest, distance to each of the L candidates) -> tmp[n][L]; per-block partial
//                     potentials; the LAST workgroup to finish adds them in block order and picks the candidate
//   k_sk_update       closest <- tmp[:, best]; block sums of closest; the LAST workgroup walks the cumulative sums (block
//                     sums, then the cells of the block) for the next centre's L targets
// Sums are in a fixed order (bit-reproducible); they are not numpy's order, so a target within rounding of a cell's boundary
// may pick the neighbouring cell: ~1e-13 n per draw (tests: the same centres as sklearn on 2e4 cells).  d <= 64, L <= 32.
constexpr int SK_LMAX = 32;
struct SkState {            // device-resident, one per seeding
  double pot;               // current potential
  int64_t cand[SK_LMAX];    // candidate cells of the centre being chosen
  int best;                 // the winning candidate
  unsigned done_a, done_b;  // arrival counters of the two kernels
};

// Workgroup = SBLK consecutive cells in sub-tiles of 128 rows, staged through LDS with coalesced loads (row stride 65: odd,
// conflict-free column walks).  Thread t works on row t & 127 and on HALF of the candidates (waves 0-1 the first half,
// waves 2-3 the second): the candidates' coordinates are wave-uniform global reads (scalar loads), the row's come from LDS,
// eight running sums per pass.  tmp is candidate-major (tmp[l][i]): coalesced writes here, one contiguous column read later.
__global__ __launch_bounds__(256) void k_sk_candidates(const double* __restrict__ x, int64_t n, int d, int L,
                                                       const double* __restrict__ closest, double* __restrict__ tmp,
                                                       double* __restrict__ part, SkState* st, const double* __restrict__ cb,
                                                       double* __restrict__ centers, int64_t* __restrict__ indices, int c) {
  constexpr int XS = 65;
  __shared__ double xs[128 * XS];
  __shared__ double cst[64 * SK_LMAX + 8];      // candidates, coordinate-major: cst[k][l]  (zero beyond L: the unused lanes of a pass)
  __shared__ double red[4][16];
  __shared__ double red8[8][SK_LMAX];
  __shared__ bool last;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int row = t & 127, half = __builtin_amdgcn_readfirstlane(t >> 7);     // (wave-uniform: scalar loads of the candidates)
  const int Lh = (L + 1) / 2, l_beg = half * Lh, l_cnt = (L - l_beg < Lh) ? (L - l_beg) : Lh;     // this thread's candidates
  double acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.0;
  for (int e = threadIdx.x; e < 64 * SK_LMAX + 8; e += 256) cst[e] = 0.0;
  __syncthreads();
  for (int e = threadIdx.x; e < L * d; e += 256) { const int l = e / d, k = e - l * d; cst[k * SK_LMAX + l] = cb[e]; }
  const int64_t base = (int64_t)blockIdx.x * SBLK;
  // A sub-tile (128 rows x d doubles) is ONE contiguous range of the row-major matrix: 32 unconditional-address loads per
  // thread, issued back to back (and for sub-tile s + 1 before the arithmetic of s), scattered into the padded LDS rows after.
  const unsigned magic = (unsigned)((0x100000000ull + (unsigned)d - 1u) / (unsigned)d);     // e / d for e < 2^16
  double stage[32];
  auto request = [&](int sub) {
    const int64_t r0 = base + sub * 128;
    const int64_t left = (n - r0) * (int64_t)d;
    const int cnt = (int)((left < 128 * (int64_t)d) ? (left > 0 ? left : 0) : 128 * d);
    const double* src = x + r0 * (int64_t)d;
#pragma unroll
    for (int q = 0; q < 32; ++q) { const int e = t + 256 * q; stage[q] = (e < cnt) ? src[e] : 0.0; }
  };
  request(0);
  for (int sub = 0; sub < SBLK / 128; ++sub) {
    const int64_t r0 = base + sub * 128;
    if (r0 >= n) break;
    const int nrows = (int)((n - r0 < 128) ? (n - r0) : 128);
    const int cnt = nrows * d;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const unsigned e = (unsigned)(t + 256 * q);
      if ((int)e < cnt) { const unsigned r = (unsigned)(((unsigned long long)e * magic) >> 32); xs[r * XS + (e - r * (unsigned)d)] = stage[q]; }
    }
    __syncthreads();
    if (sub + 1 < SBLK / 128) request(sub + 1);
    if (row < nrows) {
      const int64_t i = r0 + row;
      const double cl = closest[i];
      const double* xr = xs + row * XS;
      for (int j0 = 0; j0 < l_cnt; j0 += 8) {
        double sacc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        const double* cq = cst + l_beg + j0;             // eight candidates side by side per coordinate: broadcast LDS reads
        for (int k = 0; k < d; ++k) {
          const double v = xr[k];
#pragma unroll
          for (int j = 0; j < 8; ++j) { const double a = v - cq[k * SK_LMAX + j]; sacc[j] = fma(a, a, sacc[j]); }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j0 + j < l_cnt) {
            const double r = fmin(cl, sacc[j]);
            tmp[(int64_t)(l_beg + j0 + j) * n + i] = r;
            acc[j0 + j] += r;
          }
      }
    }
  }
  // block partials: fixed shuffle tree per wave; a candidate's two waves added in order
  for (int j = 0; j < Lh; ++j) {
    double v = acc[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[wave][j] = v;
  }
  __syncthreads();
  if (t < L) { const int h = t / Lh, j = t - h * Lh; part[(int64_t)blockIdx.x * SK_LMAX + t] = red[2 * h][j] + red[2 * h + 1][j]; }
  __threadfence();
  __syncthreads();
  if (t == 0) last = atomicAdd(&st->done_a, 1u) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  // the last workgroup: potentials of the L candidates -- eight runs of blocks per candidate, each added in block order,
  // the runs in order -- and the smallest one wins (the first on ties, as numpy.argmin)
  {
    const int l = t & 31, run = t >> 5;
    const int64_t nb = gridDim.x, per = (nb + 7) / 8, b0 = run * per, b1 = (b0 + per < nb) ? b0 + per : nb;
    double p = 0.0;
    if (l < L) for (int64_t bb = b0; bb < b1; ++bb) p += part[bb * SK_LMAX + l];
    red8[run][l] = p;
  }
  __syncthreads();
  if (t == 0) {
    int best = 0;
    double pbest = 0.0;
    for (int l = 0; l < L; ++l) {
      double p = 0.0;
      for (int r = 0; r < 8; ++r) p += red8[r][l];
      if (l == 0 || p < pbest) { best = l; pbest = p; }
    }
    st->best = best;
    st->pot = pbest;
    st->done_a = 0;
    indices[c] = st->cand[best];
  }
  __syncthreads();
  const int64_t chosen = st->cand[st->best];
  for (int k = t; k < d; k += 256) centers[(int64_t)c * d + k] = x[chosen * (int64_t)d + k];
}

// first: closest = |x - centre 0|^2 (tmp unused); else closest <- tmp[best][:].  Block sums -> bsum; the last workgroup
// draws the NEXT centre's candidates from `uni` (L uniform numbers; nullptr: none left) and copies their coordinates to cb
__global__ __launch_bounds__(256) void k_sk_update(const double* __restrict__ x, int64_t n, int d, int L,
                                                   double* __restrict__ closest, const double* __restrict__ tmp,
                                                   double* __restrict__ bsum, SkState* st, const double* __restrict__ uni,
                                                   double* __restrict__ cb, int first, int64_t first_id) {
  __shared__ double cs[64];
  __shared__ double red[4];
  __shared__ bool last;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (first) {
    for (int k = t; k < d; k += 256) cs[k] = x[first_id * (int64_t)d + k];
    __syncthreads();
  }
  const int best = first ? 0 : st->best;
  const int64_t base = (int64_t)blockIdx.x * SBLK;
  double acc = 0.0;
  for (int q = 0; q < SBLK / 256; ++q) {
    const int64_t i = base + q * 256 + t;
    if (i < n) {
      double v;
      if (first) {
        v = 0.0;
        for (int k = 0; k < d; ++k) { const double a = x[i * (int64_t)d + k] - cs[k]; v = fma(a, a, v); }
      } else {
        v = tmp[(int64_t)best * n + i];
      }
      closest[i] = v;
      acc += v;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (t == 0) bsum[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  __threadfence();
  __syncthreads();
  if (t == 0) last = atomicAdd(&st->done_b, 1u) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (t == 0) st->done_b = 0;
  if (!uni) return;
  // numpy.searchsorted(cumsum(closest), target): the cell whose interval of the running sum holds the target.  Two levels --
  // runs of block sums, then the cells of one block -- each a wave-wide scan (no workgroup barrier): wave w draws the
  // targets w, w + 4, ...; all of it in a fixed order.
  auto wave_find = [&](double v, double target, int* own, double* rem_out) {
    double inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const double o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
    const double prev = __shfl_up(inc, 1, 64);
    const double excl = (lane == 0) ? 0.0 : prev;
    const bool mine = v > 0.0 && !(target < excl) && target < inc;
    const unsigned long long mask = __ballot(mine);
    if (mask == 0ull) { *own = -1; *rem_out = 0.0; return inc; }
    const int o = __ffsll((long long)mask) - 1;
    *own = o;
    *rem_out = __shfl(target - excl, o, 64);
    return inc;
  };
  const int64_t nblk = gridDim.x, per = (nblk + 63) / 64;
  const int64_t b0 = lane * per, b1 = (b0 + per < nblk) ? b0 + per : nblk;
  // the block sums through LDS when they fit (one coalesced read instead of chains of dependent global loads)
  __shared__ double bs_lds[4096];
  const bool in_lds = nblk <= 4096;
  if (in_lds) { for (int64_t b = t; b < nblk; b += 256) bs_lds[b] = bsum[b]; }
  __syncthreads();
  const double* bs = in_lds ? bs_lds : bsum;
  double ps = 0.0;
  for (int64_t b = b0; b < b1; ++b) ps += bs[b];
  int own = 0;
  double rem = 0.0;
  const double pot = __shfl(wave_find(ps, -1.0, &own, &rem), 63, 64);          // (the scan's last inclusive sum: the total)
  if (first && t == 0) st->pot = pot;
  const double cur_pot = first ? pot : st->pot;
  for (int l = wave; l < L; l += 4) {
    (void)wave_find(ps, uni[l] * cur_pot, &own, &rem);
    double target = rem;
    if (own < 0) { own = 63; while (own > 0 && !((int64_t)own * per < nblk)) --own; target = INFINITY; }
    int64_t blk = (int64_t)own * per;
    {
      const int64_t bend = (blk + per < nblk) ? blk + per : nblk;
      for (; blk + 1 < bend; ++blk) { const double bv = bs[blk]; if (target < bv) break; target -= bv; }
    }
    const int64_t lo = blk * SBLK, hi = (lo + SBLK < n) ? lo + SBLK : n;
    constexpr int CPL = SBLK / 64;
    double v[CPL], ls = 0.0;
#pragma unroll
    for (int e = 0; e < CPL; ++e) { const int64_t i = lo + lane * CPL + e; v[e] = (i < hi) ? closest[i] : 0.0; ls += v[e]; }
    int own2 = 0;
    double rem2 = 0.0;
    (void)wave_find(ls, target, &own2, &rem2);
    int64_t pick = (hi == n) ? n - 1 : hi - 1;           // (np.clip / rounding at the end of the sums: the last cell)
    if (own2 >= 0) {
      int64_t mine = lo + lane * CPL + CPL - 1;
      double run = 0.0;
      bool found = false;
#pragma unroll
      for (int e = 0; e < CPL; ++e) { run += v[e]; if (!found && run >= rem2) { mine = lo + lane * CPL + e; found = true; } }
      if (mine >= hi) mine = hi - 1;
      pick = __shfl(mine, own2, 64);
    }
    if (lane == 0) st->cand[l] = pick;
    if (lane < d) cb[l * d + lane] = x[pick * (int64_t)d + lane];
  }
}

// centres seeded exactly as sklearn's _kmeans_plusplus would with the given random numbers: dc (m x d, device) <- the centres,
// indices (m, device) <- the cells they are
static int sklearn_seed(mln_ctx* ctx, const double* dx, int64_t n, int d, int64_t m, int64_t first_id, const double* uniforms,
                        int L, double* dc, int64_t* indices) {
  hipStream_t st = ctx->stream;
  const int64_t nblk = (n + SBLK - 1) / SBLK;
  double *closest = nullptr, *tmp = nullptr, *part = nullptr, *bsum = nullptr, *duni = nullptr, *cb = nullptr;
  SkState* state = nullptr;
  int rc = MLN_OK;
  auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == MLN_OK) rc = mln_hip_fail(ctx, e, "kmeans (sklearn seeding)", __FILE__, __LINE__); };
  chk(mln_dmalloc((void**)&closest, sizeof(double) * (size_t)n));
  chk(mln_dmalloc((void**)&tmp, sizeof(double) * (size_t)n * (size_t)L));
  chk(mln_dmalloc((void**)&part, sizeof(double) * (size_t)nblk * SK_LMAX));
  chk(mln_dmalloc((void**)&bsum, sizeof(double) * (size_t)nblk));
  chk(mln_dmalloc((void**)&duni, sizeof(double) * (size_t)std::max<int64_t>(1, (m - 1) * L)));
  chk(mln_dmalloc((void**)&state, sizeof(SkState)));
  chk(mln_dmalloc((void**)&cb, sizeof(double) * SK_LMAX * 64));
  if (rc == MLN_OK) {
    chk(hipMemsetAsync(state, 0, sizeof(SkState), st));
    if (m > 1) chk(hipMemcpyAsync(duni, uniforms, sizeof(double) * (size_t)((m - 1) * L), hipMemcpyDefault, st));
    chk(hipMemcpyAsync(dc, dx + first_id * d, sizeof(double) * (size_t)d, hipMemcpyDeviceToDevice, st));
    chk(hipMemcpyAsync(indices, &first_id, sizeof(int64_t), hipMemcpyHostToDevice, st));
    chk(hipStreamSynchronize(st));                 // (first_id is a stack variable)
  }
  if (rc == MLN_OK) {
    hipLaunchKernelGGL(k_sk_update, dim3((unsigned)nblk), dim3(256), 0, st, dx, n, d, L, closest, tmp, bsum, state, m > 1 ? duni : nullptr, cb, 1, first_id);
    for (int64_t c = 1; c < m; ++c) {
      hipLaunchKernelGGL(k_sk_candidates, dim3((unsigned)nblk), dim3(256), 0, st, dx, n, d, L, closest, tmp, part, state, cb, dc, indices, (int)c);
      hipLaunchKernelGGL(k_sk_update, dim3((unsigned)nblk), dim3(256), 0, st, dx, n, d, L, closest, tmp, bsum, state,
                         c + 1 < m ? duni + c * L : nullptr, cb, 0, (int64_t)0);
    }
    chk(hipGetLastError());
    chk(hipStreamSynchronize(st));
  }
  if (rc == MLN_OK) {
    // a NaN / inf coordinate makes the potential non-finite, every target then fails its interval test and the seeding would
    // silently pick last cells (with max_iter = 0 kmeans_level's own check never runs): refuse, like mln_kmeans
    SkState h;
    chk(hipMemcpy(&h, state, sizeof(SkState), hipMemcpyDeviceToHost));
    if (rc == MLN_OK && !std::isfinite(h.pot)) { mln_set_error(ctx, "kmeans (sklearn seeding): x contains non-finite values"); rc = MLN_ERR_ARG; }
  }
  void* ptrs[] = {closest, tmp, part, bsum, duni, state, cb};
  for (void* p : ptrs) if (p) (void)mln_dfree(p);
  return rc;
}

extern "C" int mln_kmeans_sklearn(mln_ctx* ctx, const double* x, int64_t n, int32_t d, int64_t m, int64_t first_id,
                                  const double* uniforms, int32_t n_local_trials, int32_t max_iter, double tol,
                                  double* centers, int64_t* indices_out, int32_t* n_iter_out, double* inertia_out) {
  if (!ctx || !x || !centers || (m > 1 && !uniforms)) return MLN_ERR_ARG;
  if (n < 1 || d < 1 || d > 64 || m < 1 || m > n || first_id < 0 || first_id >= n || n_local_trials < 1 || n_local_trials > SK_LMAX) {
    mln_set_error(ctx, "kmeans (sklearn seeding): bad shape (need 1 <= m <= n, d <= 64, 1 <= trials <= 32, 0 <= first < n)");
    return MLN_ERR_SHAPE;
  }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  const double* dx = x;
  double* owned = nullptr;
  {
    hipPointerAttribute_t attr;
    if (!(hipPointerGetAttributes(&attr, x) == hipSuccess && (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged))) {
      (void)hipGetLastError();
      MLN_HIP(ctx, mln_dmalloc((void**)&owned, sizeof(double) * (size_t)n * d));
      const hipError_t ce = hipMemcpyAsync(owned, x, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice, ctx->stream);
      if (ce != hipSuccess) { (void)mln_dfree(owned); return mln_hip_fail(ctx, ce, "kmeans (sklearn seeding): upload", __FILE__, __LINE__); }
      dx = owned;
    }
  }
  double* dc = nullptr;
  int64_t* dind = nullptr;
  int rc = MLN_OK;
  if (mln_dmalloc((void**)&dc, sizeof(double) * (size_t)m * d) != hipSuccess ||
      mln_dmalloc((void**)&dind, sizeof(int64_t) * (size_t)m) != hipSuccess) {
    (void)hipGetLastError();
    mln_set_error(ctx, "kmeans (sklearn seeding): out of device memory");
    rc = MLN_ERR_HIP;                    // (falls through to the clean-up below: `owned` may be an n x d upload)
  }
  if (rc == MLN_OK) rc = sklearn_seed(ctx, dx, n, d, m, first_id, uniforms, n_local_trials, dc, dind);
  if (rc == MLN_OK && indices_out && hipMemcpy(indices_out, dind, sizeof(int64_t) * (size_t)m, hipMemcpyDefault) != hipSuccess) rc = MLN_ERR_HIP;
  // Lloyd's sweeps from these centres over ALL cells, to sklearn's stopping rule (max_iter = 0: the seeding alone)
  if (rc == MLN_OK && max_iter > 0) rc = kmeans_level(ctx, dx, n, d, m, 0, max_iter, tol, dc, centers, n_iter_out, inertia_out);
  else if (rc == MLN_OK) {
    if (hipMemcpy(centers, dc, sizeof(double) * (size_t)m * d, hipMemcpyDefault) != hipSuccess) rc = MLN_ERR_HIP;
    if (n_iter_out) *n_iter_out = 0;
  }
  (void)hipStreamSynchronize(ctx->stream);
  if (dc) (void)mln_dfree(dc);
  if (dind) (void)mln_dfree(dind);
  if (owned) (void)mln_dfree(owned);
  return rc;
}

// A few Lloyd sweeps from given centres (device, m x d): the coarse clustering of the pruned 1-NN search (rowmin_f16.hip), where
// any partition is valid and k-means++'s m dependent draws would cost more than the sweeps.
int kmeans_lloyd_from(mln_ctx* ctx, const double* x, int64_t n, int32_t d, int64_t m, int32_t max_iter, const double* init, double* centers) {
  int32_t it = 0;
  return kmeans_level(ctx, x, n, d, m, 0, max_iter, 0.0, init, centers, &it, nullptr);
}

// Reference: parameters.compute_landmarks -> sklearn.cluster.k_means(x, n_landmarks, n_init=1, random_state) (parameters.py:243-291).
// Round 4, coarse to fine: with many cells per centre, Lloyd's ~200 sweeps over ALL cells mostly move centres that a
// fraction of the cells already places well.  Above 64 cells per centre (and 2e5 cells) the seeding and a first Lloyd run
// to the same tolerance use every s-th cell (max(32 m, n / 8) of them), and the full data set only polishes from there.
// Same contract as before -- k-means++ / Lloyd to sklearn's tolerance, quality checked by the inertia -- at 1e6 x 50 -> 5000
// centres 0.77 -> ~0.3 s.  n_iter_out: sweeps of the coarse run + sweeps over all cells.
extern "C" int mln_kmeans(mln_ctx* ctx, const double* x, int64_t n, int32_t d, int64_t m, int64_t seed,
                          int32_t max_iter, double tol, double* centers, int32_t* n_iter_out, double* inertia_out) {
  if (!ctx || !x || !centers) return MLN_ERR_ARG;
  if (n < 1 || d < 1 || d > 64 || m < 1 || m > n) { mln_set_error(ctx, "kmeans: bad shape (need 1 <= m <= n, d <= 64)"); return MLN_ERR_SHAPE; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  const bool two_level = n >= 64 * m && n >= 200000 && m >= 2 &&
                         !(mln_experiment("MELLON_AMD_KM_LEVELS") && std::atoi(mln_experiment("MELLON_AMD_KM_LEVELS")) == 1);
  if (!two_level) return kmeans_level(ctx, x, n, d, m, seed, max_iter, tol, nullptr, centers, n_iter_out, inertia_out);
  hipStream_t st = ctx->stream;
  // the cells on the device once, for both levels
  const double* dx = x;
  double* owned = nullptr;
  hipPointerAttribute_t attr;
  if (!(hipPointerGetAttributes(&attr, x) == hipSuccess && (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged))) {
    (void)hipGetLastError();
    MLN_HIP(ctx, mln_dmalloc((void**)&owned, sizeof(double) * (size_t)n * d));
    MLN_HIP(ctx, hipMemcpyAsync(owned, x, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice, st));
    dx = owned;
  }
  const int64_t want = std::max<int64_t>(32 * m, n / 8);
  const int64_t stride = std::max<int64_t>(1, n / want);
  const int64_t ns = (n + stride - 1) / stride;
  double *xs = nullptr, *c0 = nullptr;
  int rc = MLN_OK;
  if (mln_dmalloc((void**)&xs, sizeof(double) * (size_t)ns * d) != hipSuccess ||
      mln_dmalloc((void**)&c0, sizeof(double) * (size_t)m * d) != hipSuccess) rc = MLN_ERR_HIP;
  if (rc == MLN_OK && hipMemcpy2DAsync(xs, sizeof(double) * d, dx, sizeof(double) * d * stride, sizeof(double) * d, (size_t)ns,
                                       hipMemcpyDeviceToDevice, st) != hipSuccess) rc = MLN_ERR_HIP;
  int32_t it0 = 0, it1 = 0;
  if (rc == MLN_OK) rc = kmeans_level(ctx, xs, ns, d, m, seed, max_iter, tol, nullptr, c0, &it0, nullptr);
  if (rc == MLN_OK) rc = kmeans_level(ctx, dx, n, d, m, seed, max_iter, tol, c0, centers, &it1, inertia_out);
  if (n_iter_out) *n_iter_out = it0 + it1;
  (void)hipStreamSynchronize(st);
  if (xs) (void)mln_dfree_synced(xs);
  if (c0) (void)mln_dfree_synced(c0);
  if (owned) (void)mln_dfree_synced(owned);
  return rc;
}
