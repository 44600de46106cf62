// Row-wise nearest candidate on the fp16 matrix cores, with a rigorous error bound: the pre-filter of the exact 1-NN
// search (parameters.py:352-433: the reference's nearest-neighbour distances) and the assignment step of k-means
// (parameters.py:243-291).
//
// The exact search is 2 n m d fp64 flops on the fp64 matrix pipe (78 TFLOP/s: 2.3 s at 1e6 x 1e6 x 50).  The fp16
// pipe is 30x faster, and a double split into two halves, x = hi + lo (hi = half(x), lo = half(x - hi)), carries 21-22
// bits: the three products hi.hi + hi.lo + lo.hi, accumulated in fp32, give  s_ij = |y_j|^2 - 2 x_i.y_j  with an
// error that can be BOUNDED, per row, by a small multiple of 2^-18 |x_i| max_j |y_j|.  One sweep keeps, per query row,
// the smallest and the second-smallest approximate value and the arg of the smallest.  The caller then evaluates the
// winner exactly in fp64; if the runner-up's approximate value minus the bound still exceeds it, no other candidate
// can be closer -- the row is CERTIFIED and its distance is the exact fp64 one.  The few rows that are not (2nd
// neighbour within the bound: ~1 % at C3) are re-searched by the exact fp64 kernel.  The result is the exact
// nearest-neighbour distance for every row; only the time changes.
//
// Kernel shape: 8 waves x 32 query rows per workgroup; every wave keeps the MFMA A operands of its rows (4 k-steps x
// {hi, lo}) in registers for the whole sweep; candidate tiles (128 rows x 64 k x {hi, lo} halves = 32 KB) go through LDS
// once per workgroup, double buffered, row pitch 272 B (conflict-free 16-byte reads).  v_mfma_f32_32x32x16_f16: both
// operands are "k-contiguous rows" read with the same lane -> k map, so the k order inside the instruction is irrelevant.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mln_internal.h"

hipError_t mln_dfree_synced(void* p);   // alloc.hip: release after the caller synchronised the only stream that used p
#include "rowmin_f16.h"
// rowmin_w64.hip: the folded sweep, one wave per SIMD (round 6)
int launch_rowmin_w64(mln_ctx* ctx, const _Float16* X, int64_t n, const _Float16* Y, int64_t m, int64_t self_offset, int exclude_self,
                      float* m1, float* m2, int* arg, const int* row_idx, const uint32_t* stage_mask, int mask_words, const int* wg_order,
                      const int* n_dev = nullptr, float* smin = nullptr, int64_t smin_stride = 0);
#include "mln_options.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int KP = 64;               // padded feature count (d <= 64)
constexpr int ROWH = 2 * KP;         // halves per split row: hi[0..63] | lo[0..63]
constexpr int RT = 128;              // candidates per stage
constexpr int PITCH = 272;           // LDS bytes per candidate row (256 + 16)

// x (n x d doubles) -> split rows (n x 128 halves, zero padded), squared norms (fp64) and their fp32 roundings.
// role 1 (query rows): the coordinates are stored times -2 (exact in binary) and the three spare k slots d, d+1, d+2 of
// the hi half hold 1;  role 2 (candidate rows): those slots hold |y|^2 as three halves (hi + mid + lo, 33 bits) -- the
// matrix product of a query and a candidate row is then  |y|^2 - 2 x.y  itself, no epilogue arithmetic.  Needs d <= 61.
__global__ __launch_bounds__(256) void k_split_f16(const double* __restrict__ x, int64_t n, int d, _Float16* __restrict__ out,
                                                   double* __restrict__ xx, float* __restrict__ xxf, int role,
                                                   const double* __restrict__ prep) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int k = threadIdx.x & 63;
  if (row >= n) return;
  // prep (rowmin_prepare): [0..63] the centre, [64] the power-of-two scale -- the split holds (x - centre) * scale, whose
  // squared norms stay below 2^14: nothing overflows half precision whatever the units or the offset of the data
  const double v0 = (k < d) ? (x[row * d + k] - prep[k]) * prep[64] : 0.0;
  double s = v0 * v0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  double v = (role == 1) ? -2.0 * v0 : v0;
  _Float16 hi = (_Float16)v;
  _Float16 lo = (_Float16)(v - (double)hi);
  if (role == 1 && k >= d && k < d + 3) { hi = (_Float16)1.0f; lo = (_Float16)0.0f; }
  if (role == 2 && k >= d && k < d + 3) {
    const _Float16 n0 = (_Float16)s;
    const _Float16 n1 = (_Float16)(s - (double)n0);
    const _Float16 n2 = (_Float16)(s - (double)n0 - (double)n1);
    hi = (k == d) ? n0 : (k == d + 1 ? n1 : n2);
    lo = (_Float16)0.0f;
  }
  out[row * ROWH + k] = hi;
  out[row * ROWH + KP + k] = lo;
  if (k == 0) { if (xx) xx[row] = s; if (xxf) xxf[row] = (float)s; }
}

// Centre of the data: the mean of up to 4096 evenly spaced rows, summed in a fixed order (deterministic).  One workgroup.
__global__ __launch_bounds__(256) void k_sample_centre(const double* __restrict__ y, int64_t m, int d, double* __restrict__ prep) {
  __shared__ double part[4][64];
  const int k = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t cnt = m < 4096 ? m : 4096;
  const int64_t stride = m / cnt;
  double s = 0.0;
  if (k < d)
    for (int64_t r = g; r < cnt; r += 4) s += y[(r * stride) * d + k];
  part[g][k] = s;
  __syncthreads();
  if (g == 0) prep[k] = (k < d) ? (part[0][k] + part[1][k] + part[2][k] + part[3][k]) / (double)cnt : 0.0;
  if (threadIdx.x == 0) { prep[64] = 1.0; prep[65] = 0.0; }
}

// prep[65] = max over rows of |x - centre|^2 (a maximum: order-independent, so the atomic is deterministic;
// non-negative doubles order like their bit patterns)
__global__ __launch_bounds__(256) void k_max_centred_norm(const double* __restrict__ x, int64_t n, int d, double* __restrict__ prep) {
  __shared__ double red[4];
  double mx = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double s = 0.0;
    for (int k = 0; k < d; ++k) { const double t = x[i * d + k] - prep[k]; s = fma(t, t, s); }
    mx = fmax(mx, s);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double v = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    atomicMax(reinterpret_cast<unsigned long long*>(prep + 65), (unsigned long long)__double_as_longlong(v));
  }
}

// one workgroup: out = max xx; several (out zeroed by the launcher): an integer maximum of the bit patterns (non-negative doubles)
__global__ void k_max_norm(const double* __restrict__ xx, int64_t n, double* __restrict__ out) {
  __shared__ double red[4];
  double m = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmax(m, xx[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double v = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    if (gridDim.x == 1) out[0] = v;
    else atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(v));
  }
}

// min of two floats as ONE v_min_f32.  fminf() is llvm.minnum, and in IEEE mode the compiler has to quiet a possible
// signalling NaN first: a v_max_f32 x, x in front of every operand it cannot prove clean -- here the MFMA accumulator of
// every element and the running minimum, i.e. four vector instructions per element where the epilogue is meant to be two
// (round 5, counters: 5.5 vector instructions per element, matrix cores busy 0.48 of the sweep; worth 3 % of the sweep).  The hardware instruction
// itself returns the other operand for a quiet NaN, which is all an MFMA can produce.
__device__ __forceinline__ float vmin_raw(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// The sweep for PLAIN split operands (d > 61: no spare k slots for the folded |y|^2 and -2; the folded sweep is
// rowmin_w64.hip).  TOP2: also the second-smallest value per row (1-NN certification); else only the arg of the smallest
// (k-means labels).
template <bool TOP2>
__global__ __launch_bounds__(512) void k_rowmin_f16x3(const _Float16* __restrict__ Xs, int64_t n,
                                                      const _Float16* __restrict__ Ys, int64_t m,
                                                      const float* __restrict__ yyf, int64_t self_offset, int exclude_self,
                                                      float* __restrict__ out_m1, float* __restrict__ out_m2,
                                                      int* __restrict__ out_arg, const int* __restrict__ row_idx) {
  extern __shared__ unsigned char lds[];                      // 2 x (RT x PITCH) candidate rows + 2 x RT norms
  float* ynl = reinterpret_cast<float*>(lds + 2 * RT * PITCH);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lg = lane >> 5;
  const int64_t row0w = (int64_t)blockIdx.x * 256 + wave * 32;
  // A operands: this wave's 32 rows, k = 16 ks + 8 lg + e
  h8 ahi[4], alo[4];
  {
    // row_idx (k-means with bounds: the rows whose bounds no longer decide): query row r of this launch is row
    // row_idx[r] of Xs; the outputs stay in launch order
    const int64_t ar0 = (row0w + lr < n) ? row0w + lr : n - 1;
    const int64_t ar = row_idx ? (int64_t)row_idx[ar0] : ar0;
    const _Float16* src = Xs + ar * ROWH + 8 * lg;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      ahi[ks] = *reinterpret_cast<const h8*>(src + 16 * ks);
      alo[ks] = *reinterpret_cast<const h8*>(src + KP + 16 * ks);
    }
  }
  float m1[16], m2[16];
  int a1[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { m1[r] = INFINITY; m2[r] = INFINITY; a1[r] = 0; }

  // staging: 128 rows x 256 B = 2048 16-byte pieces per stage, 4 per thread
  v4i st[4];
  float stn = 0.f;
  auto g_load = [&](int64_t col0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int piece = tid + 512 * q, r = piece >> 4, seg = piece & 15;
      const int64_t c = col0 + r;
      st[q] = (c < m) ? *reinterpret_cast<const v4i*>(Ys + c * ROWH + seg * 8) : v4i{0, 0, 0, 0};
    }
    if (tid < RT) stn = (col0 + tid < m) ? yyf[col0 + tid] : INFINITY;
  };
  auto l_store = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int piece = tid + 512 * q, r = piece >> 4, seg = piece & 15;
      *reinterpret_cast<v4i*>(lds + buf * (RT * PITCH) + r * PITCH + seg * 16) = st[q];
    }
    if (tid < RT) ynl[buf * RT + tid] = stn;
  };
  g_load(0);
  l_store(0);
  __syncthreads();
  int buf = 0;
  for (int64_t col0 = 0; col0 < m; col0 += RT, buf ^= 1) {
    const bool more = col0 + RT < m;
    if (more) g_load(col0 + RT);
    const unsigned char* base = lds + buf * (RT * PITCH);
#pragma unroll
    for (int sub = 0; sub < RT / 32; ++sub) {
      f16v hh, cx;
#pragma unroll
      for (int r = 0; r < 16; ++r) { hh[r] = 0.f; cx[r] = 0.f; }
      const unsigned char* brow = base + (sub * 32 + lr) * PITCH + 16 * lg;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const h8 bhi = *reinterpret_cast<const h8*>(brow + 32 * ks);
        const h8 blo = *reinterpret_cast<const h8*>(brow + 2 * KP + 32 * ks);
        hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], bhi, hh, 0, 0, 0);
        cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], blo, cx, 0, 0, 0);
        cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[ks], bhi, cx, 0, 0, 0);
      }
      const float yc = ynl[buf * RT + sub * 32 + lr];
      const int col = (int)(col0 + sub * 32 + lr);
      // the excluded pair (i, i + self_offset) can only sit in a tile that meets this wave's diagonal band
      const int64_t lo_c = col0 + sub * 32, d0 = row0w + self_offset;
      const bool diag = exclude_self && lo_c < d0 + 32 && lo_c + 32 > d0;
      if (!diag) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float s = fmaf(-2.f, hh[r] + cx[r], yc);
          if (TOP2) m2[r] = fminf(m2[r], fmaxf(m1[r], s));
          a1[r] = (s < m1[r]) ? col : a1[r];
          m1[r] = fminf(m1[r], s);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = row0w + (r & 3) + 8 * (r >> 2) + 4 * lg;
          float s = fmaf(-2.f, hh[r] + cx[r], yc);
          if ((int64_t)col == row + self_offset) s = INFINITY;
          if (TOP2) m2[r] = fminf(m2[r], fmaxf(m1[r], s));
          a1[r] = (s < m1[r]) ? col : a1[r];
          m1[r] = fminf(m1[r], s);
        }
      }
    }
    if (more) l_store(buf ^ 1);
    __syncthreads();
  }
  // merge the 32 column-lanes of each row (lanes with the same lg hold the same rows)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      const float o1 = __shfl_xor(m1[r], off, 64), o2 = __shfl_xor(m2[r], off, 64);
      const int oa = __shfl_xor(a1[r], off, 64);
      if (TOP2) m2[r] = fminf(fmaxf(m1[r], o1), fminf(m2[r], o2));
      // ties go to the smaller candidate index: the result does not depend on the lane order of the merge
      const bool take = (o1 < m1[r]) || (o1 == m1[r] && oa < a1[r]);
      a1[r] = take ? oa : a1[r];
      m1[r] = fminf(m1[r], o1);
    }
    const int64_t row = row0w + (r & 3) + 8 * (r >> 2) + 4 * lg;
    if (lr == 0 && row < n) {
      out_m1[row] = m1[r];
      if (TOP2) out_m2[row] = m2[r];
      out_arg[row] = a1[r];
    }
  }
}

// Bound on |s~_j - s_j| for every candidate j of a row with |x| = xn (centred, scaled units) when max_j |y_j| = yn: see the
// comment in k_nn_certify.
__device__ __forceinline__ double rowmin_value_bound(double xn, double yn) {
  const double u = 5.9604644775390625e-08;   // 2^-24
  const double mag = 2.0 * xn * yn + yn * yn;
  return 1.25 * (1.9073486328125e-06 * 2.0 * xn * yn + 80.0 * u * mag + 160.0 * u * 9.765625e-04 * mag +
                 4.0 * u * mag + 64.0 * u * 2.0 * (2.0 * xn + yn) + 1e-300);
}

// Exact fp64 value of the winner, certification against the runner-up, list of the rows that need the exact search.
//   s_j = |y_j|^2 - 2 x.y_j (exact);  |s~_j - s_j| <= E_i for every j  =>  j* != arg implies s_{j*} >= m2~ - E_i.
// fold > 0: the winner is one of arg + 32 q, q < fold (rowmin_fold_candidates()): all of them are evaluated exactly.
__global__ __launch_bounds__(256) void k_nn_certify(const double* __restrict__ x, int64_t n, const double* __restrict__ y, int64_t m, int d,
                                                    const double* __restrict__ xx, const double* __restrict__ yy,
                                                    const float* __restrict__ m2, const int* __restrict__ arg,
                                                    const double* __restrict__ yy_max, int fold, int64_t self_offset,
                                                    const double* __restrict__ prep,
                                                    double* __restrict__ out, int* __restrict__ n_flag, int* __restrict__ flagged,
                                                    float* __restrict__ fthr, double* __restrict__ fdd) {
  // eight lanes per row, every eighth coordinate each (a wave's load covers 8 consecutive rows: one thread per row walked its
  // 400-byte row alone -- 7.6 ms for 1e6 rows where the data are 2 GB), partial sums added in a fixed order
  const int sub = threadIdx.x & 7;
  const int64_t i = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
  if (i >= n) return;                                  // (whole groups of eight leave together)
  // everything the sweep saw is (x - centre) * scale: xx, yy, m2 and the bound E live in those units; the winner's value
  // s is formed from the same centred and scaled coordinates in fp64
  const double sc = prep[64];
  double s = INFINITY;
  int64_t js = -1;
  for (int q = 0; q < (fold ? fold : 1); ++q) {        // fold: the number of candidates a folded arg stands for (0: exact column)
    const int64_t j = (int64_t)arg[i] + 32 * q;
    if (j >= m || j == i + self_offset) continue;
    double dot = 0.0;
    for (int k = sub; k < d; k += 8) dot = fma((x[i * d + k] - prep[k]) * sc, (y[j * d + k] - prep[k]) * sc, dot);
    dot += __shfl_xor(dot, 1, 64);
    dot += __shfl_xor(dot, 2, 64);
    dot += __shfl_xor(dot, 4, 64);
    const double sv = yy[j] - 2.0 * dot;
    if (sv < s) { s = sv; js = j; }
  }
  const double xn = sqrt(xx[i]), yn = sqrt(yy_max[0]);
  // The value the sweep compares is |y|^2 - 2 x.y from three half-precision products (hi.hi + hi.lo + lo.hi) accumulated in
  // fp32.  Dropped terms of the split: 2^-19 of 2 |x| |y|; fp32 accumulation: <= 17 roundings per MFMA x 4 MFMAs + slack
  // = 80 u on the magnitudes of the hi.hi terms (2 |x| |y| + |y|^2), 160 u on the 2^-10 smaller cross terms; |y|^2 in
  // fp32 (plain) or as three halves (fold); half-precision subnormals of tiny coordinates (2^-24 each, 64 of them).
  const double E = rowmin_value_bound(xn, yn);
  const bool certified = ((double)m2[i] - E) > s;
  // reported: the winner's distance from its coordinates (sum (x_k - y_k)^2), not from the cancelling |x|^2 - 2 x.y + |y|^2:
  // exact 0 for a duplicated cell, relative error ~eps otherwise (see nn_direct_distance in cov_kernels.hip)
  // (one lane, the coordinates in order: the sum k_nn_list_eval and the exact search report -- a row's value does not depend
  //  on which of them resolved it; both rows were just read by the eight lanes)
  if (sub != 0) return;
  double dd = INFINITY;
  if (js >= 0) {
    dd = 0.0;
    for (int k = 0; k < d; ++k) {
      const double t = x[i * d + k] - y[js * d + k];
      dd = fma(t, t, dd);
    }
  }
  out[i] = sqrt(dd);
  if (!certified) {
    const int slot = atomicAdd(n_flag, 1);
    flagged[slot] = (int)i;
    // for the candidate list of k_rowmin_list: a candidate that beats the winner has s_j <= s, hence s~_j <= s + E; the
    // threshold is that, rounded UP to fp32 (two ulps of slack), and the winner's squared distance goes along
    if (fthr) {
      float t = (float)(s + E);
      t = nextafterf(nextafterf(t, INFINITY), INFINITY);
      fthr[slot] = (js >= 0) ? t : INFINITY;
      fdd[slot] = dd;
    }
  }
}

// Second sweep over the rows the certification left open (row_idx, cnt of them): instead of the smallest two values, every
// candidate whose approximate value lies below the row's threshold -- the only ones that can be nearer than the winner --
// is appended to a list of (row slot, candidate) pairs; k_nn_list_eval then takes the exact minimum over each row's pairs.
// The fp64 search over ALL candidates that used to resolve these rows (5 % of them at C3: 112 ms of a 0.5 s search) becomes
// a second fp16 sweep over 5 % of the rows plus a few exact distances per row.  FOLD operands only (d <= 61).
// Round 6: a workgroup holds LIST_ROWS = 64 row slots (the open rows of a pruned search come ~40 to a group: with 256-row
// tiles 85 % of the products were padding -- 28 ms of a 100 ms search); its eight waves are two row halves x four quarters
// of a stage's 128 candidates.  The candidate blocks of a group are dealt to the gridDim.y workgroups of its row by ORDINAL
// among the blocks its mask selects (every y-th one): contiguous column segments left most of them idle, a group's
// candidates being a few contiguous runs of the cluster-sorted order.
constexpr int LIST_ROWS = 64;
__global__ __launch_bounds__(512) void k_rowmin_list(const _Float16* __restrict__ Xs, int64_t cnt, const int* __restrict__ row_idx,
                                                     const _Float16* __restrict__ Ys, int64_t m, const float* __restrict__ fthr,
                                                     int* __restrict__ n_pairs, int cap, int* __restrict__ pair_row,
                                                     int* __restrict__ pair_col, const uint32_t* __restrict__ stage_mask, int mask_words) {
  extern __shared__ unsigned char lds[];
  __shared__ int stop_s[2];     // (two slots by stage parity: the slot a stage reads is not written again before its barrier)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { stop_s[0] = 0; stop_s[1] = 0; }
  const int lr = lane & 31, lg = lane >> 5;
  const int quarter = wave >> 1;                                    // which 32 of a stage's 128 candidates
  const int64_t row0w = (int64_t)blockIdx.x * LIST_ROWS + (wave & 1) * 32;
  h8 ahi[4], alo[4];
  {
    const int64_t ar0 = (row0w + lr < cnt) ? row0w + lr : cnt - 1;
    const int rid = row_idx[ar0];                                   // (< 0: a padding slot -- its threshold is -inf)
    const _Float16* src = Xs + (int64_t)(rid < 0 ? 0 : rid) * ROWH + 8 * lg;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      ahi[ks] = *reinterpret_cast<const h8*>(src + 16 * ks);
      alo[ks] = *reinterpret_cast<const h8*>(src + KP + 16 * ks);
    }
  }
  float thr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = row0w + (r & 3) + 8 * (r >> 2) + 4 * lg;
    thr[r] = (row < cnt) ? fthr[row] : -INFINITY;
  }
  v4i st[4];
  auto g_load = [&](int64_t col0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int piece = tid + 512 * q, r = piece >> 4, seg = piece & 15;
      const int64_t c = col0 + r;
      st[q] = (c < m) ? *reinterpret_cast<const v4i*>(Ys + c * ROWH + seg * 8) : v4i{0, 0, 0, 0};
    }
  };
  auto l_store = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int piece = tid + 512 * q, r = piece >> 4, seg = piece & 15;
      *reinterpret_cast<v4i*>(lds + buf * (RT * PITCH) + r * PITCH + seg * 16) = st[q];
    }
  };
  // stage_mask (the pruned search): bit b of this workgroup's row selects the candidates [256 b, 256 b + 256) -- two stages here
  // (copied to LDS first: a global load per stage would sit on the critical path)
  uint32_t* mrow = stage_mask ? reinterpret_cast<uint32_t*>(lds + 2 * RT * PITCH) : nullptr;
  if (mrow) {
    for (int w = tid; w < mask_words; w += 512) mrow[w] = stage_mask[(int64_t)blockIdx.x * mask_words + w];
    __syncthreads();
  }
  // this workgroup's candidate blocks: the selected ones whose ordinal is blockIdx.y modulo gridDim.y (wave-uniform scalar work)
  const int64_t nb256 = (m + 255) / 256;
  const int ny = (int)gridDim.y, yy = (int)blockIdx.y;
  int64_t scan_blk = 0;         // the next block to look at
  int skip = yy;                // selected blocks to pass over before this workgroup's next one
  auto next_block = [&]() -> int64_t {                       // this workgroup's next block; nb256 when none is left
    if (!mrow) {
      const int64_t b = scan_blk + skip;
      scan_blk = b + 1; skip = ny - 1;
      return b < nb256 ? b : nb256;
    }
    while (scan_blk < nb256) {
      // (the word as UNSIGNED before the shift: readfirstlane returns an int, whose arithmetic shift filled the top with copies
      //  of bit 31 -- phantom selected blocks that put the workgroups' ordinals out of step, ~10 rows in 1e6 lost their nearest
      //  neighbour to the runner-up)
      const uint32_t word = (uint32_t)__builtin_amdgcn_readfirstlane((int)mrow[scan_blk >> 5]);
      const uint32_t w = word >> (uint32_t)(scan_blk & 31);
      const int have = __builtin_popcount(w);
      if (have <= skip) { skip -= have; scan_blk = ((scan_blk >> 5) + 1) * 32; continue; }      // (a whole word passed over)
      uint32_t rest = w;
      for (int k = 0; k < skip; ++k) rest &= rest - 1;       // drop the `skip` lowest set bits
      const int64_t b = scan_blk + __builtin_ctz(rest);
      scan_blk = b + 1; skip = ny - 1;
      return b;
    }
    return nb256;
  };
  // the stages: both halves of a block (the second only if it holds candidates), then the next block
  int64_t blk = next_block();
  int64_t col0 = blk < nb256 ? blk * 256 : m;
  auto stage_after = [&](int64_t c) -> int64_t {             // the stage that follows column c; m when none
    if ((c & 255) == 0 && c + RT < m) return c + RT;
    blk = next_block();
    return blk < nb256 ? blk * 256 : m;
  };
  if (col0 < m) { g_load(col0); l_store(0); }
  __syncthreads();
  int buf = 0, stage_no = 0;
  for (; col0 < m; buf ^= 1) {
    // A list that has outgrown its buffer is abandoned by the caller (exact search instead): stop feeding it.  Without this a
    // tight cluster of 50 000 mutual near-ties appends 2.5e9 pairs -- seconds of atomics on one address, and a 32-bit counter
    // that wraps to negative slots (found by tools/robustness_sweep_large.py as a write fault).
    // (Thread 0 reads the counter before the barrier that ends a stage; everybody acts on that one value after it.)
    if (stop_s[buf]) return;
    const int64_t following = stage_after(col0);
    const bool more = following < m;
    if (more) g_load(following);
    const unsigned char* brow = lds + buf * (RT * PITCH) + (quarter * 32 + lr) * PITCH + 16 * lg;
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    h8 bhi[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {          // (the same order of products as the sweep: the same values)
      bhi[ks] = *reinterpret_cast<const h8*>(brow + 32 * ks);
      const h8 blo = *reinterpret_cast<const h8*>(brow + 2 * KP + 32 * ks);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], blo, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[ks], bhi[ks], acc, 0, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[ks], bhi[ks], acc, 0, 0, 0);
    {
      const int64_t col = col0 + quarter * 32 + lr;
      // one comparison per element; the append is the rare path
      bool any = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) any = any || (acc[r] < thr[r]);
      if (any && col < m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (acc[r] < thr[r]) {
            const int slot = atomicAdd(n_pairs, 1);
            if (slot >= 0 && slot < cap) { pair_row[slot] = (int)(row0w + (r & 3) + 8 * (r >> 2) + 4 * lg); pair_col[slot] = (int)col; }
          }
        }
      }
    }
    if (more) l_store(buf ^ 1);
    // (every eighth stage: the counter's L2 round trip sat in front of every barrier -- 8 us per stage where the products take 3)
    if (tid == 0 && (++stage_no & 7) == 0) {
      const int over = __hip_atomic_load(n_pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > cap ? 1 : 0;
      stop_s[0] = over; stop_s[1] = over;
    }
    __syncthreads();
    col0 = following;
  }
}

// exact squared distance of every listed pair, minimum per row slot (non-negative doubles order like their bit patterns)
__global__ __launch_bounds__(256) void k_nn_list_eval(const double* __restrict__ x, const double* __restrict__ y, int d,
                                                      const int* __restrict__ flagged, const int* __restrict__ pair_row,
                                                      const int* __restrict__ pair_col, int n_pairs, int64_t self_offset,
                                                      unsigned long long* __restrict__ best) {
  // (one thread per pair, the coordinates in order: the same sum the certification and the exact search report)
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n_pairs) return;
  const int rs = pair_row[p];
  const int64_t i = flagged[rs], j = pair_col[p];
  if (j == i + self_offset) return;
  const double* xr = x + i * d;
  const double* yr = y + j * d;
  double dd = 0.0;
  for (int k = 0; k < d; ++k) { const double t = xr[k] - yr[k]; dd = fma(t, t, dd); }
  atomicMin(&best[rs], (unsigned long long)__double_as_longlong(dd));
}

__global__ void k_nn_list_finish(const int* __restrict__ flagged, int cnt, const unsigned long long* __restrict__ best,
                                 const double* __restrict__ fdd, double* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= cnt || flagged[r] < 0) return;          // (negative: a padding slot of the list sweep)
  const double b = __longlong_as_double((long long)best[r]);
  out[flagged[r]] = sqrt(fmin(b, fdd[r]));
}

// labels from the fold variant's stage-level args: the closest of the ncand candidates, in fp64
__global__ __launch_bounds__(256) void k_resolve_labels(const double* __restrict__ x, int64_t n, const double* __restrict__ y, int64_t m,
                                                        int d, const double* __restrict__ yy, int* __restrict__ arg, int ncand) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double best = INFINITY;
  int bj = arg[i];
  for (int q = 0; q < ncand; ++q) {
    const int64_t j = (int64_t)arg[i] + 32 * q;
    if (j >= m) continue;
    double dot = 0.0;
    for (int k = 0; k < d; ++k) dot = fma(x[i * d + k], y[j * d + k], dot);
    const double sv = yy[j] - 2.0 * dot;
    if (sv < best) { best = sv; bj = (int)j; }
  }
  arg[i] = bj;
}

// k-means with distance bounds (kmeans.hip): one searched row -> its label and the two bounds.
//   the sweep (TOP2, FOLD) returned arg (the winner is one of arg + 32 q, q < ncand) and m2~, the second smallest approximate
//   value; every candidate other than the owner of the smallest has s~_j >= m2~, hence s_j >= m2~ - E.  They are
//   evaluated exactly, sum (x_k - c_k)^2 in fp64: label = the closest of them, ub = its distance, and
//   lb^2 = min(second closest of them, (|x'|^2 + m2~ - E) / scale^2) bounds the distance to EVERY other centre.
//   (The label need not be the true nearest centre when two are within E of each other; then ub > lb and the row is
//   simply searched again next sweep.)
// rows: r < cnt, i = idx ? idx[r] : r.  sums / counts (optional): the row moves from its old label to the new one.
__global__ __launch_bounds__(256) void k_km_resolve(const double* __restrict__ x, int64_t cnt, const int* __restrict__ idx,
                                                    const double* __restrict__ c, int64_t m, int d,
                                                    const double* __restrict__ xxs, const double* __restrict__ yy_max,
                                                    const double* __restrict__ prep, const float* __restrict__ m2,
                                                    const int* __restrict__ arg, int* __restrict__ label,
                                                    double* __restrict__ ub, double* __restrict__ lb,
                                                    double* __restrict__ sums, double* __restrict__ counts,
                                                    const double* __restrict__ colscale, int ncand,
                                                    const int* __restrict__ cnt_dev, const int* __restrict__ cperm,
                                                    KmGroups grp, const uint32_t* __restrict__ stage_mask, int mask_words) {
  // cnt_dev: the row count lives on the device (cnt sized the grid).
  // Group bounds (kmeans.hip): the sweep saw the centres in a permuted order -- candidate POSITION j is centre cperm[j] -- and
  // only the stages (256 positions) in the row block's stage_mask; a second sweep left the smallest value of every swept
  // stage in grp.smin.  Every swept stage's bound grp.lbg[i][s] is renewed:
  //   the winner's stage        the bound for "every candidate but the winner" (second / m2~, as without groups)
  //   any other swept stage     the larger of that and the stage's own minimum
  //   and never below what was known (the old bound; if the label changes, the old centre joins its stage's others).
  // eight lanes per row, every eighth coordinate each (coalesced over the 8 rows of a wave's load)
  const int sub = threadIdx.x & 7;
  const int64_t r = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
  if (cnt_dev) cnt = *cnt_dev;
  if (r >= cnt) return;                       // (whole groups of eight leave together: the shuffles below stay inside a group)
  const int64_t i = idx ? (int64_t)idx[r] : r;
  const double* xr = x + i * d;
  double best = INFINITY, second = INFINITY;
  int bj = arg[r], bjp = arg[r];
  for (int q = 0; q < ncand; ++q) {
    const int64_t jp = (int64_t)arg[r] + 32 * q;
    if (jp >= m) continue;
    const int64_t j = cperm ? (int64_t)cperm[jp] : jp;
    const double* cr = c + j * d;
    double dd = 0.0;
    for (int k = sub; k < d; k += 8) { const double t = xr[k] - cr[k]; dd = fma(t, t, dd); }
    dd += __shfl_xor(dd, 1, 64);
    dd += __shfl_xor(dd, 2, 64);
    dd += __shfl_xor(dd, 4, 64);
    if (dd < best) { second = best; best = dd; bj = (int)j; bjp = (int)jp; }
    else if (dd < second) second = dd;
  }
  const double sc = prep[64];
  const double E = rowmin_value_bound(sqrt(xxs[i]), sqrt(yy_max[0]));
  const double rest = fmax((xxs[i] + (double)m2[r] - E) / (sc * sc), 0.0);
  const int old = label[i];
  const double l = sqrt(fmin(second, rest));
  if (grp.lbg) {
    const double ub_old = ub[i];                               // (still the distance to the OLD centre)
    const uint32_t* mk = stage_mask ? stage_mask + (r >> 8) * mask_words : nullptr;
    const int sb = bjp >> 8, sa = grp.cpos[old] >> 8;
    for (int s = sub; s < grp.nstage; s += 8) {                // the eight lanes of the row share the stages
      if (mk && !((mk[s >> 5] >> (s & 31)) & 1u)) continue;
      double val = l;
      if (s != sb) val = fmax(l, sqrt(fmax((xxs[i] + (double)grp.smin[s * grp.smin_stride + r] - E) / (sc * sc), 0.0)));
      double eff = (double)grp.lbg[i * grp.nstage + s] - grp.cum[s];
      if (old != bj && s == sa) eff = fmin(eff, ub_old);
      grp.lbg[i * grp.nstage + s] = __double2float_rd(fmax(eff, val) + grp.cum[s]);
    }
  }
  if (sub == 0) {
    ub[i] = sqrt(best);
    lb[i] = l;
    label[i] = bj;
  }
  if (sums && old != bj) {
    // fixed-point sums (kmeans.hip, k_accumulate): the cell takes out of its old cluster exactly what it put in
    unsigned long long* isums = reinterpret_cast<unsigned long long*>(sums);
    for (int k = sub; k < d; k += 8) {
      const unsigned long long q = (unsigned long long)llrint(xr[k] * colscale[k]);
      atomicAdd(&isums[(int64_t)old * d + k], 0ULL - q);
      atomicAdd(&isums[(int64_t)bj * d + k], q);
    }
    if (sub == 0) { atomicAdd(&counts[old], -1.0); atomicAdd(&counts[bj], 1.0); }
  }
}

__global__ void k_gather_rows_excl(const double* __restrict__ x, int d, const int* __restrict__ idx, int cnt,
                                   int64_t self_offset, double* __restrict__ xg, int64_t* __restrict__ excl) {
  const int r = blockIdx.x;
  if (r >= cnt) return;
  const int64_t i = idx[r];
  for (int k = threadIdx.x; k < d; k += blockDim.x) xg[(int64_t)r * d + k] = x[i * d + k];
  if (threadIdx.x == 0) excl[r] = i + self_offset;
}

__global__ void k_scatter_rows(const double* __restrict__ vals, const int* __restrict__ idx, int cnt, double* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < cnt) out[idx[r]] = vals[r];
}


// ---- pruned exact 1-NN (round 6): coarse clusters + the triangle inequality decide which candidate BLOCKS a row block sweeps ----
// cells are sorted by the coarse cluster they were assigned to; for x in cluster a and y in cluster b,
// |x - y| >= |c_a - c_b| - r_a - r_b (r: the largest distance of a member to its centre, exact in fp64).
struct NnPrune {
  const int* label_sorted;      // n: cluster of every sorted cell
  const int64_t* offsets;       // Kc + 1: first sorted cell of every cluster
  const double* centers;        // Kc x d
  const double* radius;         // Kc
  int Kc;
};

// exact distance of every cell to the centre it was assigned to; cluster radii (max) and sizes
__global__ __launch_bounds__(256) void k_prune_assign_stats(const double* __restrict__ x, int64_t n, int d, const double* __restrict__ c,
                                                            const int* __restrict__ label, unsigned long long* __restrict__ radius_bits,
                                                            int* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int l = label[i];
  double dd = 0.0;
  for (int k = 0; k < d; ++k) { const double t = x[i * d + k] - c[(int64_t)l * d + k]; dd = fma(t, t, dd); }
  // (rounded UP a little: the radius must not be smaller than the true distance)
  const double r = sqrt(dd) * (1.0 + 1e-12);
  atomicMax(&radius_bits[l], (unsigned long long)__double_as_longlong(r));
  atomicAdd(&counts[l], 1);
}

__global__ void k_prune_offsets(const int* __restrict__ counts, int Kc, int64_t* __restrict__ offsets, int* __restrict__ cursor) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int64_t s = 0;
    for (int k = 0; k < Kc; ++k) { offsets[k] = s; s += counts[k]; cursor[k] = 0; }
    offsets[Kc] = s;
  }
}

// sorted position of every cell (order inside a cluster: arrival order -- the distances do not depend on it), its row copied there
__global__ __launch_bounds__(256) void k_prune_scatter(const double* __restrict__ x, int64_t n, int d, const int* __restrict__ label,
                                                       const int64_t* __restrict__ offsets, int* __restrict__ cursor,
                                                       int* __restrict__ perm, int* __restrict__ label_sorted, double* __restrict__ xsorted) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const int l = label[i];
  int pos = 0;
  if (lane == 0) pos = atomicAdd(&cursor[l], 1);
  pos = __shfl(pos, 0, 64);
  const int64_t p = offsets[l] + pos;
  if (lane == 0) { perm[p] = (int)i; label_sorted[p] = l; }
  for (int k = lane; k < d; k += 64) xsorted[p * d + k] = x[i * d + k];
}

// evenly spaced sample rows
__global__ void k_prune_sample(const double* __restrict__ x, int64_t stride, int64_t ns, int d, double* __restrict__ out) {
  const int64_t r = blockIdx.x;
  if (r >= ns) return;
  for (int k = threadIdx.x; k < d; k += blockDim.x) out[r * d + k] = x[r * stride * d + k];
}

// bit B of row B: every row block sweeps its own 256 cells first (the upper bounds)
__global__ void k_prune_mask_own(uint32_t* __restrict__ mask, int nblk, int words) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nblk) mask[(int64_t)b * words + (b >> 5)] = 1u << (b & 31);      // (the matrix was zeroed)
}

// per row block of 256 sorted cells: the largest upper bound of a member's nearest-neighbour distance, in ORIGINAL units:
// the best approximate value s~ = |y|^2 - 2 x.y inside the block satisfies s <= s~ + E, so dist^2 <= |x|^2 + s~ + E
__global__ __launch_bounds__(256) void k_prune_block_bound(const double* __restrict__ xx, const float* __restrict__ m1, int64_t n,
                                                           const double* __restrict__ yy_max, const double* __restrict__ prep,
                                                           double* __restrict__ bub) {
  __shared__ double red[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double v = 0.0;
  if (i < n) {
    const double E = rowmin_value_bound(sqrt(xx[i]), sqrt(yy_max[0]));
    const double s = (double)m1[i];
    v = isfinite(s) ? fmax(xx[i] + s + E, 0.0) : INFINITY;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double sc = prep[64];
    const double m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    bub[blockIdx.x] = sqrt(m) / sc * (1.0 + 1e-9);
  }
}

__global__ void k_prune_sqnorms(const double* __restrict__ x, int64_t n, int d, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int k = 0; k < d; ++k) s = fma(x[i * d + k], x[i * d + k], s);
  out[i] = s;
}

// dmin[B][b] = the smallest distance of a cell of row block B to centre b, from G = X C^T (fp64 matrix cores):
// |x - c|^2 = |x|^2 - 2 x.c + |c|^2.  For y in cluster b and x in block B: |x - y| >= |x - c_b| - |y - c_b| >= dmin[B][b] - r_b --
// in many dimensions far sharper than |c_a - c_b| - r_a - r_b (the cells of a cluster sit on a shell AROUND its centre).
__global__ __launch_bounds__(256) void k_prune_block_dmin(const double* __restrict__ G, int Kc, const double* __restrict__ xxo,
                                                          const double* __restrict__ cc, int64_t n, double* __restrict__ dmin) {
  const int B = blockIdx.x;
  const int64_t r_lo = (int64_t)B * 256, r_hi = r_lo + 256 < n ? r_lo + 256 : n;
  for (int b = threadIdx.x; b < Kc; b += 256) {
    double best = INFINITY;
    const double c2 = cc[b];
    // (rounded DOWN per cell: the three-term form cancels with an error of ~50 eps (|x|^2 + |c|^2) including the product's own)
    for (int64_t i = r_lo; i < r_hi; ++i) {
      const double xi = xxo[i];
      best = fmin(best, (xi - 2.0 * G[i * Kc + b] + c2) - 1e-13 * (xi + c2));
    }
    dmin[(int64_t)B * Kc + b] = sqrt(fmax(best, 0.0)) * (1.0 - 1e-12);
  }
}

// the candidate blocks of row block B: every block that holds a cell of a cluster b with dmin[B][b] - r_b < bound(B).
// One workgroup per row block, a thread per cluster b.
__global__ __launch_bounds__(256) void k_prune_mask(NnPrune pr, const double* __restrict__ dmin, int64_t n, const double* __restrict__ bub,
                                                    uint32_t* __restrict__ mask, int words, int* __restrict__ n_stages,
                                                    int* __restrict__ block_stages) {
  extern __shared__ uint32_t bits[];
  const int B = blockIdx.x;
  for (int w = threadIdx.x; w < words; w += 256) bits[w] = 0u;
  __syncthreads();
  const double bound = bub[B];
  for (int b = threadIdx.x; b < pr.Kc; b += 256) {
    const int64_t o0 = pr.offsets[b], o1 = pr.offsets[b + 1];
    if (o1 <= o0) continue;
    const bool cand = !isfinite(bound) || dmin[(int64_t)B * pr.Kc + b] - pr.radius[b] < bound;
    if (cand)
      for (int64_t s = o0 >> 8; s <= (o1 - 1) >> 8; ++s) atomicOr(&bits[s >> 5], 1u << (s & 31));
  }
  __syncthreads();
  int cnt = 0;
  for (int w = threadIdx.x; w < words; w += 256) { mask[(int64_t)B * words + w] = bits[w]; cnt += __popc(bits[w]); }
  if (cnt) { atomicAdd(n_stages, cnt); atomicAdd(&block_stages[B], cnt); }
}

// the list sweep's workgroup g handles the open rows flagged[64 g .. 64 g + 64) (ascending): the union of their row blocks' masks
__global__ __launch_bounds__(256) void k_prune_union_mask(const int* __restrict__ flagged, int cnt, const uint32_t* __restrict__ mask, int words,
                                                          uint32_t* __restrict__ out) {
  const int g = blockIdx.x;
  const int lo = g * LIST_ROWS, hi = lo + LIST_ROWS < cnt ? lo + LIST_ROWS : cnt;
  for (int w = threadIdx.x; w < words; w += 256) {
    uint32_t u = 0u;
    int last = -1;
    for (int sidx = lo; sidx < hi; ++sidx) {
      const int f = flagged[sidx];
      if (f < 0) continue;                                           // padding slot
      const int B = f >> 8;
      if (B != last) { u |= mask[(int64_t)B * words + w]; last = B; }
    }
    out[(int64_t)g * words + w] = u;
  }
}

__global__ void k_prune_unsort(const double* __restrict__ vals, const int* __restrict__ perm, int64_t n, double* __restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) out[perm[p]] = vals[p];
}

}  // namespace

size_t rowmin_split_bytes(int64_t rows) { return sizeof(_Float16) * (size_t)rows * ROWH; }

// Centre and scale for the half-precision copies of y (m x d) and, if given, x (n x d): prep[0..63] = the mean of an
// evenly spaced sample of y, prep[64] = the power of two that brings the largest centred squared norm of either set into
// [2^12, 2^14) (1 for all-equal data).  Distances do not depend on the centre, and a power-of-two scale is exact: only the
// RANGE of what half precision has to hold changes -- raw counts in the thousands or coordinates of 1e-6 neither
// overflow nor flush to zero.  prep: ROWMIN_PREP_DOUBLES doubles on the device.
int rowmin_prepare(mln_ctx* ctx, const double* y, int64_t m, const double* x, int64_t n, int d, double* prep) {
  if (m <= 0) return MLN_OK;
  hipLaunchKernelGGL(k_sample_centre, dim3(1), dim3(256), 0, ctx->stream, y, m, d, prep);
  hipLaunchKernelGGL(k_max_centred_norm, dim3(1024), dim3(256), 0, ctx->stream, y, m, d, prep);
  if (x && n > 0 && x != y) hipLaunchKernelGGL(k_max_centred_norm, dim3(1024), dim3(256), 0, ctx->stream, x, n, d, prep);
  double mx = 0.0;
  MLN_HIP(ctx, hipMemcpyAsync(&mx, prep + 65, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double scale = 1.0;
  if (std::isfinite(mx) && mx > 0.0) {
    int e = 0;
    (void)std::frexp(mx, &e);                                   // mx = f * 2^e, f in [0.5, 1)
    const int q = 14 - e;
    scale = std::ldexp(1.0, q >= 0 ? q / 2 : -((1 - q) / 2));    // floor(q / 2): mx * scale^2 in [2^12, 2^14)
  }
  MLN_HIP(ctx, hipMemcpyAsync(prep + 64, &scale, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));      // (scale lives on this stack frame)
  return MLN_OK;
}

int launch_split_f16(mln_ctx* ctx, const double* x, int64_t n, int d, void* split, double* xx, float* xxf, int role,
                     const double* prep) {
  if (n <= 0) return MLN_OK;
  if (d > KP || (role != 0 && d > KP - 3)) { mln_set_error(ctx, "split_f16: too many features"); return MLN_ERR_UNSUPPORTED; }
  hipLaunchKernelGGL(k_split_f16, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, x, n, d,
                     reinterpret_cast<_Float16*>(split), xx, xxf, role, prep);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

// A folded sweep (rowmin_w64.hip) reports, with the runner-up tracked, the winner's column only per half stage (128
// candidates) and lane: it is one of arg + 32 q, q < rowmin_fold_candidates().
int rowmin_fold_candidates() { return 4; }

int launch_rowmin_f16x3(mln_ctx* ctx, const void* xs, int64_t n, const void* ys, int64_t m, const float* yyf,
                        int64_t self_offset, int exclude_self, float* m1, float* m2, int* arg, int fold, const int* row_idx) {
  if (n <= 0 || m <= 0) return MLN_OK;
  if (m > 2147483647LL) { mln_set_error(ctx, "rowmin: too many candidates"); return MLN_ERR_UNSUPPORTED; }
  const _Float16* X = reinterpret_cast<const _Float16*>(xs);
  const _Float16* Y = reinterpret_cast<const _Float16*>(ys);
  if (fold) return launch_rowmin_w64(ctx, X, n, Y, m, self_offset, exclude_self, m1, m2, arg, row_idx, nullptr, 0, nullptr);
  const size_t lds_bytes = (size_t)2 * RT * PITCH + 2 * RT * sizeof(float);
  static bool attr = false;
  if (!attr) {
    MLN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rowmin_f16x3<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    MLN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_rowmin_f16x3<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr = true;
  }
  const dim3 grid((unsigned)((n + 255) / 256)), block(512);
  if (m2) hipLaunchKernelGGL((k_rowmin_f16x3<true>), grid, block, lds_bytes, ctx->stream, X, n, Y, m, yyf, self_offset, exclude_self, m1, m2, arg, row_idx);
  else hipLaunchKernelGGL((k_rowmin_f16x3<false>), grid, block, lds_bytes, ctx->stream, X, n, Y, m, yyf, self_offset, exclude_self, m1, nullptr, arg, row_idx);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

// the folded sweep restricted per 256-row workgroup to the candidate blocks its row of `stage_mask` selects (kmeans.hip)
int launch_rowmin_masked(mln_ctx* ctx, const void* xs, int64_t n_max, const int* n_dev, const void* ys, int64_t m, float* m1, float* m2,
                         int* arg, const int* row_idx, const uint32_t* stage_mask, int mask_words, float* smin, int64_t smin_stride) {
  if (n_max <= 0 || m <= 0) return MLN_OK;
  return launch_rowmin_w64(ctx, reinterpret_cast<const _Float16*>(xs), n_max, reinterpret_cast<const _Float16*>(ys), m, 0, 0, m1, m2, arg,
                           row_idx, stage_mask, mask_words, nullptr, n_dev, smin, smin_stride);
}

int launch_resolve_labels(mln_ctx* ctx, const double* x, int64_t n, const double* y, int64_t m, int d, const double* yy, int* arg) {
  if (n <= 0) return MLN_OK;
  hipLaunchKernelGGL(k_resolve_labels, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, x, n, y, m, d, yy, arg, rowmin_fold_candidates());
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_km_resolve(mln_ctx* ctx, const double* x, int64_t cnt, const int* idx, const double* c, int64_t m, int d,
                      const double* xxs, const double* yy_max, const double* prep, const float* m2, const int* arg,
                      int* label, double* ub, double* lb, double* sums, double* counts, const double* colscale,
                      const int* cnt_dev, const int* cperm, const KmGroups* grp, const uint32_t* stage_mask, int mask_words) {
  if (cnt <= 0) return MLN_OK;
  KmGroups g{};
  if (grp) g = *grp;
  hipLaunchKernelGGL(k_km_resolve, dim3((unsigned)((cnt + 31) / 32)), dim3(256), 0, ctx->stream, x, cnt, idx, c, m, d, xxs,
                     yy_max, prep, m2, arg, label, ub, lb, sums, counts, colscale, rowmin_fold_candidates(), cnt_dev, cperm, g,
                     stage_mask, mask_words);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

// the first group bounds, after the full sweeps: lbg[i][s] = the bound for every centre but the label (lb) in the label's stage,
// the larger of that and the stage's minimum elsewhere (grp.cum is zero)
__global__ void k_km_init_groups(int64_t n, const int* __restrict__ label, const double* __restrict__ lb,
                                 const double* __restrict__ xxs, const double* __restrict__ yy_max, const double* __restrict__ prep,
                                 KmGroups grp) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * grp.nstage) return;
  const int64_t i = t / grp.nstage;
  const int s = (int)(t - i * grp.nstage);
  const double sc = prep[64];
  const double E = rowmin_value_bound(sqrt(xxs[i]), sqrt(yy_max[0]));
  double val = lb[i];
  if (s != (grp.cpos[label[i]] >> 8)) val = fmax(val, sqrt(fmax((xxs[i] + (double)grp.smin[s * grp.smin_stride + i] - E) / (sc * sc), 0.0)));
  grp.lbg[t] = __double2float_rd(val);
}
int launch_km_init_groups(mln_ctx* ctx, int64_t n, const int* label, const double* lb, const double* xxs, const double* yy_max,
                          const double* prep, const KmGroups* grp) {
  hipLaunchKernelGGL(k_km_init_groups, dim3((unsigned)((n * grp->nstage + 255) / 256)), dim3(256), 0, ctx->stream, n, label, lb, xxs,
                     yy_max, prep, *grp);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_max_norm(mln_ctx* ctx, const double* xx, int64_t n, double* out) {
  const unsigned grid = n > 65536 ? 512u : 1u;                 // (one workgroup over 1e6 values: 1.6 ms of the 1-NN search)
  if (grid > 1) MLN_HIP(ctx, hipMemsetAsync(out, 0, sizeof(double), ctx->stream));
  hipLaunchKernelGGL(k_max_norm, dim3(grid), dim3(256), 0, ctx->stream, xx, n, out);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

// Exact 1-NN distances through the fp16 pre-filter (see the head of this file).  x: n x d, y: m x d (device), d <= 64.
// stats (optional, host): [0] rows re-searched exactly.
static int nn_search_core(mln_ctx* ctx, const double* x, int64_t n, const double* y, int64_t m, int d,
                          int64_t self_offset, double* out, double* stats, const NnPrune* prune) {
  const bool same = (x == y && n == m);
  void *xs = nullptr, *ys = nullptr;
  double *xx = nullptr, *yy = nullptr, *ymax = nullptr, *prep = nullptr;
  float *yyf = nullptr, *m1 = nullptr, *m2 = nullptr;
  int *arg = nullptr, *nflag = nullptr, *flagged = nullptr;
  std::vector<void*> owned;
  auto alloc = [&](void** p, size_t bytes) -> bool {
    if (mln_dmalloc(p, bytes > 0 ? bytes : 8) != hipSuccess) return false;
    owned.push_back(*p);
    return true;
  };
  auto cleanup = [&](int rc) {
    (void)hipStreamSynchronize(ctx->stream);
    for (void* p : owned) (void)mln_dfree_synced(p);
    return rc;
  };
  const int fold = (d <= KP - 3) ? 1 : 0;       // (three spare k slots carry |y|^2)
  const bool share = same && !fold;             // without folding, queries and candidates are the same split copy
  bool ok = alloc(&xs, rowmin_split_bytes(n)) && alloc((void**)&xx, sizeof(double) * n) &&
            alloc((void**)&m1, sizeof(float) * n) && alloc((void**)&m2, sizeof(float) * n) &&
            alloc((void**)&arg, sizeof(int) * n) && alloc((void**)&nflag, sizeof(int)) &&
            alloc((void**)&flagged, sizeof(int) * n) && alloc((void**)&ymax, sizeof(double)) &&
            alloc((void**)&yyf, sizeof(float) * m) && alloc((void**)&prep, sizeof(double) * ROWMIN_PREP_DOUBLES);
  if (ok && !share) ok = alloc(&ys, rowmin_split_bytes(m));
  if (ok && !same) ok = alloc((void**)&yy, sizeof(double) * m);
  if (!ok) { mln_set_error(ctx, "nn_distances: out of device memory"); return cleanup(MLN_ERR_HIP); }
  int rc = rowmin_prepare(ctx, y, m, same ? nullptr : x, n, d, prep);
  if (rc == MLN_OK) rc = launch_split_f16(ctx, x, n, d, xs, xx, share ? yyf : nullptr, fold ? 1 : 0, prep);
  if (rc == MLN_OK && !share) rc = launch_split_f16(ctx, y, m, d, ys, same ? nullptr : yy, yyf, fold ? 2 : 0, prep);
  if (share) ys = xs;
  if (same) yy = xx;
  if (rc != MLN_OK) return cleanup(rc);
  if (launch_max_norm(ctx, yy, m, ymax) != MLN_OK) return cleanup(MLN_ERR_HIP);
  if (hipMemsetAsync(nflag, 0, sizeof(int), ctx->stream) != hipSuccess) return cleanup(MLN_ERR_HIP);
  float* fthr = nullptr;
  double* fdd = nullptr;
  if (fold && (!alloc((void**)&fthr, sizeof(float) * n) || !alloc((void**)&fdd, sizeof(double) * n))) {
    mln_set_error(ctx, "nn_distances: out of device memory"); return cleanup(MLN_ERR_HIP);
  }
  const bool list_ok = !(mln_experiment("MELLON_AMD_NN_LIST") && std::atoi(mln_experiment("MELLON_AMD_NN_LIST")) == 0);
  // (Measured and taken out, round 5: a first sweep with the hi.hi product alone -- a third of the matrix work, values to
  //  2^-9 |x||y| instead of 2^-18.  At C3 it left 55 % of the rows open -- in 50 dimensions the gap to the second neighbour
  //  is a few per cent of |x||y| -- and took 249 ms against the three-product sweep's 360: its two-instruction epilogue is
  //  as long as its MFMAs.)
  int cnt = 0;
  const uint32_t* pr_mask = nullptr;      // the pruned search's candidate-block masks (one row per 256-row block)
  int pr_words = 0;
  if (prune && fold && same) {
    // pass 1: every row block against its own 256 cells -> an upper bound of each member's nearest-neighbour distance;
    // pass 2: the candidate blocks the triangle inequality cannot exclude for that bound (k_prune_mask)
    const int nblk = (int)((n + 255) / 256), words = (nblk + 31) / 32;
    uint32_t* mask = nullptr;
    double* bub = nullptr;
    int *nst = nullptr, *wg_order = nullptr;
    if (!alloc((void**)&mask, sizeof(uint32_t) * (size_t)nblk * words) || !alloc((void**)&bub, sizeof(double) * nblk) ||
        !alloc((void**)&nst, sizeof(int))) { mln_set_error(ctx, "nn_distances: out of device memory"); return cleanup(MLN_ERR_HIP); }
    if (hipMemsetAsync(mask, 0, sizeof(uint32_t) * (size_t)nblk * words, ctx->stream) != hipSuccess ||
        hipMemsetAsync(nst, 0, sizeof(int), ctx->stream) != hipSuccess) return cleanup(MLN_ERR_HIP);
    hipLaunchKernelGGL(k_prune_mask_own, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, ctx->stream, mask, nblk, words);
    rc = launch_rowmin_w64(ctx, reinterpret_cast<const _Float16*>(xs), n, reinterpret_cast<const _Float16*>(ys), m, self_offset, 1, m1,
                           nullptr, arg, nullptr, mask, words, nullptr);
    if (rc != MLN_OK) return cleanup(rc);
    hipLaunchKernelGGL(k_prune_block_bound, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, xx, m1, n, ymax, prep, bub);
    {
      const int Kc = prune->Kc;
      double *G = nullptr, *dmin = nullptr, *xxo = nullptr, *cc = nullptr;
      if (!alloc((void**)&G, sizeof(double) * (size_t)n * Kc) || !alloc((void**)&dmin, sizeof(double) * (size_t)nblk * Kc) ||
          !alloc((void**)&xxo, sizeof(double) * n) || !alloc((void**)&cc, sizeof(double) * Kc)) {
        mln_set_error(ctx, "nn_distances: out of device memory"); return cleanup(MLN_ERR_HIP);
      }
      hipLaunchKernelGGL(k_prune_sqnorms, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, x, n, d, xxo);
      hipLaunchKernelGGL(k_prune_sqnorms, dim3((unsigned)((Kc + 255) / 256)), dim3(256), 0, ctx->stream, prune->centers, (int64_t)Kc, d, cc);
      GemmArgs g{};
      g.A = x; g.lda = d; g.B = prune->centers; g.ldb = d; g.C = G; g.ldc = Kc;
      g.M = n; g.N = Kc; g.K = d; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 1;
      rc = launch_dgemm(ctx, g);
      if (rc != MLN_OK) return cleanup(rc);
      hipLaunchKernelGGL(k_prune_block_dmin, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, G, Kc, xxo, cc, n, dmin);
      int* bst = nullptr;
      if (!alloc((void**)&bst, sizeof(int) * nblk) || !alloc((void**)&wg_order, sizeof(int) * nblk)) { mln_set_error(ctx, "nn_distances: out of device memory"); return cleanup(MLN_ERR_HIP); }
      if (hipMemsetAsync(bst, 0, sizeof(int) * nblk, ctx->stream) != hipSuccess) return cleanup(MLN_ERR_HIP);
      hipLaunchKernelGGL(k_prune_mask, dim3((unsigned)nblk), dim3(256), sizeof(uint32_t) * (size_t)words, ctx->stream, *prune, dmin, n, bub,
                         mask, words, nst, bst);
      // heaviest row blocks first (a block whose bound is infinite sweeps everything: not in the last round)
      std::vector<int> hb((size_t)nblk), ord((size_t)nblk);
      if (hipMemcpyAsync(hb.data(), bst, sizeof(int) * nblk, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess) return cleanup(MLN_ERR_HIP);
      for (int i = 0; i < nblk; ++i) ord[(size_t)i] = i;
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return hb[(size_t)a] > hb[(size_t)b]; });
      if (hipMemcpyAsync(wg_order, ord.data(), sizeof(int) * nblk, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess) return cleanup(MLN_ERR_HIP);
    }
    rc = launch_rowmin_w64(ctx, reinterpret_cast<const _Float16*>(xs), n, reinterpret_cast<const _Float16*>(ys), m, self_offset, 1, m1,
                           m2, arg, nullptr, mask, words, wg_order);
    if (rc != MLN_OK) return cleanup(rc);
    pr_mask = mask; pr_words = words;
    if (std::getenv("MELLON_AMD_TRACE") || stats) {
      int hst = 0;
      if (hipMemcpyAsync(&hst, nst, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess) {
        if (stats) stats[2] = (double)hst / ((double)nblk * (double)nblk);
        if (std::getenv("MELLON_AMD_TRACE"))
          std::fprintf(stderr, "[trace] nn_distances: %d of %lld (row block, candidate block) pairs swept (%.1f %%)\n", hst,
                       (long long)nblk * nblk, 100.0 * hst / ((double)nblk * nblk));
      }
    }
  } else {
    rc = launch_rowmin_f16x3(ctx, xs, n, ys, m, yyf, self_offset, 1, m1, m2, arg, fold, nullptr);
    if (rc != MLN_OK) return cleanup(rc);
  }
  hipLaunchKernelGGL(k_nn_certify, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, ctx->stream, x, n, y, m, d, xx, yy, m2, arg,
                     ymax, fold ? rowmin_fold_candidates() : 0, self_offset, prep, out, nflag, flagged, fthr, fdd);
  if (hipMemcpyAsync(&cnt, nflag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) return cleanup(mln_hip_fail(ctx, hipGetLastError(), "nn certify", __FILE__, __LINE__));
  if (std::getenv("MELLON_AMD_TRACE"))
    std::fprintf(stderr, "[trace] nn_distances: the sweep left %d of %lld rows open\n", cnt, (long long)n);
  if (stats) stats[0] = (double)cnt;
  // The open rows: a second fp16 sweep that LISTS the candidates below each row's threshold, exact distances of the listed
  // pairs.  A list that outgrows its buffer (masses of exact duplicates) falls through to the exact search below.
  bool listed = false;
  int* flagged_compact = nullptr;
  if (cnt > 0 && fold && list_ok) {
    // Pruned search: a candidate that could beat an open row's winner lies in a block its row block's mask selects (everything
    // else is provably no nearer than a cell that block has seen).  The open rows are put in ascending order -- neighbours in
    // the sorted order share clusters -- in groups of 32 to 64 (the other slots of a group's 64 are padding with a -inf
    // threshold), and each list workgroup sweeps its share of the union of its rows' masks.
    const uint32_t* lmask = nullptr;
    int cnt_list = cnt;
    if (pr_mask && cnt <= 262144) {
      // (the compact list of open rows stays available for the exact fall-back below)
      if (!alloc((void**)&flagged_compact, sizeof(int) * (size_t)cnt)) { mln_set_error(ctx, "nn_distances: out of device memory"); return cleanup(MLN_ERR_HIP); }
      if (hipMemcpyAsync(flagged_compact, flagged, sizeof(int) * (size_t)cnt, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) return cleanup(MLN_ERR_HIP);
      std::vector<int> hf((size_t)cnt);
      std::vector<float> ht((size_t)cnt);
      std::vector<double> hd((size_t)cnt);
      if (hipMemcpyAsync(hf.data(), flagged, sizeof(int) * (size_t)cnt, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipMemcpyAsync(ht.data(), fthr, sizeof(float) * (size_t)cnt, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipMemcpyAsync(hd.data(), fdd, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess) return cleanup(MLN_ERR_HIP);
      std::vector<int> order((size_t)cnt);
      for (int i = 0; i < cnt; ++i) order[(size_t)i] = i;
      std::sort(order.begin(), order.end(), [&](int a, int b) { return hf[(size_t)a] < hf[(size_t)b]; });
      // real rows per list group: 32 to 64 (a group's tile is LIST_ROWS slots; ~240 groups x 8 column shares when the rows are few)
      const int G = std::min(LIST_ROWS, std::max(32, (cnt + 239) / 240));
      const int ngrp = (cnt + G - 1) / G;
      cnt_list = ngrp * LIST_ROWS;
      std::vector<int> sf((size_t)cnt_list, -1);
      std::vector<float> stv((size_t)cnt_list, -INFINITY);
      std::vector<double> sd((size_t)cnt_list, INFINITY);
      for (int i = 0; i < cnt; ++i) {
        const size_t slot = (size_t)(i / G) * LIST_ROWS + (size_t)(i % G), o = (size_t)order[(size_t)i];
        sf[slot] = hf[o]; stv[slot] = ht[o]; sd[slot] = hd[o];
      }
      uint32_t* um = nullptr;
      if (!alloc((void**)&um, sizeof(uint32_t) * (size_t)ngrp * pr_words)) { mln_set_error(ctx, "nn_distances: out of device memory"); return cleanup(MLN_ERR_HIP); }
      if (hipMemcpyAsync(flagged, sf.data(), sizeof(int) * (size_t)cnt_list, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
          hipMemcpyAsync(fthr, stv.data(), sizeof(float) * (size_t)cnt_list, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
          hipMemcpyAsync(fdd, sd.data(), sizeof(double) * (size_t)cnt_list, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess) return cleanup(MLN_ERR_HIP);      // (the host vectors go out of scope)
      hipLaunchKernelGGL(k_prune_union_mask, dim3((unsigned)ngrp), dim3(256), 0, ctx->stream, flagged, cnt_list, pr_mask, pr_words, um);
      lmask = um;
      if (std::getenv("MELLON_AMD_TRACE")) {
        std::vector<uint32_t> hu((size_t)ngrp * pr_words);
        if (hipMemcpyAsync(hu.data(), um, sizeof(uint32_t) * hu.size(), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
            hipStreamSynchronize(ctx->stream) == hipSuccess) {
          long long bitsum = 0;
          for (uint32_t w : hu) bitsum += __builtin_popcount(w);
          std::fprintf(stderr, "[trace] nn_distances: %d open rows in %d list workgroups of %d; their masks select %.1f %% of the candidate blocks\n",
                       cnt, ngrp, G, 100.0 * (double)bitsum / ((double)ngrp * (double)((n + 255) / 256)));
        }
      }
    }
    const int cap = (int)std::min<int64_t>((int64_t)32 * cnt + 65536, (int64_t)1 << 28);
    int *npairs = nullptr, *prow = nullptr, *pcol = nullptr;
    unsigned long long* best = nullptr;
    if (!alloc((void**)&npairs, sizeof(int)) || !alloc((void**)&prow, sizeof(int) * (size_t)cap) ||
        !alloc((void**)&pcol, sizeof(int) * (size_t)cap) || !alloc((void**)&best, sizeof(unsigned long long) * (size_t)cnt_list)) {
      mln_set_error(ctx, "nn_distances: out of device memory"); return cleanup(MLN_ERR_HIP);
    }
    if (hipMemsetAsync(npairs, 0, sizeof(int), ctx->stream) != hipSuccess ||
        hipMemsetAsync(best, 0x7f, sizeof(unsigned long long) * (size_t)cnt_list, ctx->stream) != hipSuccess) return cleanup(MLN_ERR_HIP);
    if (lmask && pr_words > 4096) lmask = nullptr;                 // (the mask row lives in LDS)
    const size_t lds_bytes = (size_t)2 * RT * PITCH + (lmask ? (size_t)pr_words * sizeof(uint32_t) : 0);
    static bool attr_l = false;
    if (!attr_l) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_rowmin_list), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)((size_t)2 * RT * PITCH + 4096 * sizeof(uint32_t))) != hipSuccess)
        return cleanup(MLN_ERR_HIP);
      attr_l = true;
    }
    // column shares: enough workgroups for every CU, and no workgroup with more than an eighth of the candidates
    const int n_groups = (cnt_list + LIST_ROWS - 1) / LIST_ROWS;
    const int n_seg = (m >= 65536) ? std::max(8, std::min(64, 4096 / std::max(1, n_groups))) : 1;
    hipLaunchKernelGGL(k_rowmin_list, dim3((unsigned)n_groups, (unsigned)n_seg), dim3(512), lds_bytes, ctx->stream,
                       reinterpret_cast<const _Float16*>(xs), (int64_t)cnt_list, flagged, reinterpret_cast<const _Float16*>(ys), m, fthr,
                       npairs, cap, prow, pcol, lmask, pr_words);
    int np = 0;
    if (hipMemcpyAsync(&np, npairs, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) return cleanup(mln_hip_fail(ctx, hipGetLastError(), "nn list", __FILE__, __LINE__));
    if (stats) stats[1] = (double)np;
    if (std::getenv("MELLON_AMD_TRACE")) std::fprintf(stderr, "[trace] nn_distances: %d candidate pairs listed for %d rows (capacity %d)\n", np, cnt, cap);
    if (np <= cap) {
      if (np > 0)
        hipLaunchKernelGGL(k_nn_list_eval, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, ctx->stream, x, y, d, flagged, prow, pcol, np,
                           self_offset, best);
      hipLaunchKernelGGL(k_nn_list_finish, dim3((unsigned)((cnt_list + 255) / 256)), dim3(256), 0, ctx->stream, flagged, cnt_list, best, fdd, out);
      listed = true;
    }
  }
  if (cnt > 0 && !listed) {
    // the uncertified rows: exact fp64 search, each with its own excluded candidate
    double *xg = nullptr, *og = nullptr;
    int64_t* excl = nullptr;
    if (!alloc((void**)&xg, sizeof(double) * (size_t)cnt * d) || !alloc((void**)&og, sizeof(double) * cnt) ||
        !alloc((void**)&excl, sizeof(int64_t) * cnt)) { mln_set_error(ctx, "nn_distances: out of device memory"); return cleanup(MLN_ERR_HIP); }
    const int* open_rows = flagged_compact ? flagged_compact : flagged;
    hipLaunchKernelGGL(k_gather_rows_excl, dim3((unsigned)cnt), dim3(64), 0, ctx->stream, x, d, open_rows, cnt, self_offset, xg, excl);
    rc = launch_nn_distances_exact(ctx, xg, cnt, y, m, d, 0, excl, og);
    if (rc != MLN_OK) return cleanup(rc);
    hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, og, open_rows, cnt, out);
  }
  if (hipGetLastError() != hipSuccess) return cleanup(MLN_ERR_HIP);
  return cleanup(MLN_OK);
}

int kmeans_lloyd_from(mln_ctx* ctx, const double* x, int64_t n, int32_t d, int64_t m, int32_t max_iter, const double* init, double* centers);   // kmeans.hip

// Exact 1-NN among the cells of ONE set (x against itself, the pair (i, i) excluded) with cluster pruning: a coarse k-means of a
// sample (Kc centres, a few sweeps: any partition is valid, a good one prunes more), the cells sorted by cluster, and the
// search of nn_search_core restricted per row block to the candidate blocks a triangle-inequality bound cannot exclude.  The
// result is the exact distance for every row, as without pruning: an excluded candidate is provably no nearer than a cell the
// row block has already seen; rows the certification leaves open are resolved against ALL candidates as before.
static int nn_distances_pruned(mln_ctx* ctx, const double* x, int64_t n, int d, double* out, double* stats) {
  int Kc = (int)std::min<int64_t>(1024, std::max<int64_t>(64, n / 2048));
  const int64_t ns = std::min<int64_t>(n, (int64_t)64 * Kc), stride = n / ns;
  std::vector<void*> owned;
  auto alloc = [&](void** p, size_t bytes) -> bool {
    if (mln_dmalloc(p, bytes > 0 ? bytes : 8) != hipSuccess) return false;
    owned.push_back(*p);
    return true;
  };
  auto cleanup = [&](int rc) {
    (void)hipStreamSynchronize(ctx->stream);
    for (void* p : owned) (void)mln_dfree_synced(p);
    return rc;
  };
  double *sample = nullptr, *cent = nullptr, *prep = nullptr, *radius = nullptr, *xsorted = nullptr, *osorted = nullptr;
  void *xs1 = nullptr, *cs = nullptr;
  float *ccf = nullptr, *m1f = nullptr;
  int *label = nullptr, *counts = nullptr, *cursor = nullptr, *perm = nullptr, *lsorted = nullptr;
  int64_t* offsets = nullptr;
  bool ok = alloc((void**)&sample, sizeof(double) * (size_t)ns * d) && alloc((void**)&cent, sizeof(double) * (size_t)Kc * d) &&
            alloc((void**)&prep, sizeof(double) * ROWMIN_PREP_DOUBLES) && alloc((void**)&radius, sizeof(double) * Kc) &&
            alloc((void**)&xsorted, sizeof(double) * (size_t)n * d) && alloc((void**)&osorted, sizeof(double) * n) &&
            alloc(&xs1, rowmin_split_bytes(n)) && alloc(&cs, rowmin_split_bytes(Kc)) && alloc((void**)&ccf, sizeof(float) * Kc) &&
            alloc((void**)&m1f, sizeof(float) * n) && alloc((void**)&label, sizeof(int) * n) && alloc((void**)&counts, sizeof(int) * Kc) &&
            alloc((void**)&cursor, sizeof(int) * Kc) && alloc((void**)&perm, sizeof(int) * n) && alloc((void**)&lsorted, sizeof(int) * n) &&
            alloc((void**)&offsets, sizeof(int64_t) * (Kc + 1));
  if (!ok) { mln_set_error(ctx, "nn_distances: out of device memory"); return cleanup(MLN_ERR_HIP); }
  hipLaunchKernelGGL(k_prune_sample, dim3((unsigned)ns), dim3(64), 0, ctx->stream, x, stride, ns, d, sample);
  // seeds: every 64th cell of the sample; four Lloyd sweeps (any partition is valid -- a better one only prunes more)
  double* seeds = nullptr;
  if (!alloc((void**)&seeds, sizeof(double) * (size_t)Kc * d)) { mln_set_error(ctx, "nn_distances: out of device memory"); return cleanup(MLN_ERR_HIP); }
  hipLaunchKernelGGL(k_prune_sample, dim3((unsigned)Kc), dim3(64), 0, ctx->stream, sample, ns / Kc, (int64_t)Kc, d, seeds);
  int rc = kmeans_lloyd_from(ctx, sample, ns, d, Kc, 4, seeds, cent);
  if (rc != MLN_OK) return cleanup(rc);
  // every cell to its (approximately) nearest centre: the folded sweep without the runner-up tracks the exact column
  rc = rowmin_prepare(ctx, x, n, nullptr, 0, d, prep);
  if (rc == MLN_OK) rc = launch_split_f16(ctx, x, n, d, xs1, nullptr, nullptr, 1, prep);
  if (rc == MLN_OK) rc = launch_split_f16(ctx, cent, Kc, d, cs, nullptr, ccf, 2, prep);
  if (rc == MLN_OK) rc = launch_rowmin_f16x3(ctx, xs1, n, cs, Kc, ccf, 0, 0, m1f, nullptr, label, 1, nullptr);
  if (rc != MLN_OK) return cleanup(rc);
  if (hipMemsetAsync(radius, 0, sizeof(double) * Kc, ctx->stream) != hipSuccess ||
      hipMemsetAsync(counts, 0, sizeof(int) * Kc, ctx->stream) != hipSuccess) return cleanup(MLN_ERR_HIP);
  hipLaunchKernelGGL(k_prune_assign_stats, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, x, n, d, cent, label,
                     reinterpret_cast<unsigned long long*>(radius), counts);
  hipLaunchKernelGGL(k_prune_offsets, dim3(1), dim3(64), 0, ctx->stream, counts, Kc, offsets, cursor);
  hipLaunchKernelGGL(k_prune_scatter, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, x, n, d, label, offsets, cursor, perm, lsorted,
                     xsorted);
  if (hipGetLastError() != hipSuccess) return cleanup(MLN_ERR_HIP);
  NnPrune pr{lsorted, offsets, cent, radius, Kc};
  rc = nn_search_core(ctx, xsorted, n, xsorted, n, d, 0, osorted, stats, &pr);
  if (rc != MLN_OK) return cleanup(rc);
  hipLaunchKernelGGL(k_prune_unsort, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, osorted, perm, n, out);
  if (hipGetLastError() != hipSuccess) return cleanup(MLN_ERR_HIP);
  return cleanup(MLN_OK);
}

int nn_distances_prefiltered(mln_ctx* ctx, const double* x, int64_t n, const double* y, int64_t m, int d,
                             int64_t self_offset, double* out, double* stats) {
  // one set against itself, folded operands, enough cells for the clustering to pay: the pruned search
  // (MELLON_AMD_NN_PRUNE=0, experiment: every row block sweeps every candidate block)
  const bool prune_ok = !(mln_experiment("MELLON_AMD_NN_PRUNE") && std::atoi(mln_experiment("MELLON_AMD_NN_PRUNE")) == 0);
  if (prune_ok && x == y && n == m && self_offset == 0 && d <= KP - 3 && n >= (int64_t)1 << 18 && n <= (int64_t)1 << 25)
    return nn_distances_pruned(ctx, x, n, d, out, stats);
  return nn_search_core(ctx, x, n, y, m, d, self_offset, out, stats, nullptr);
}
