// Row selection for the SECOND preconditioner of the MAP solve (api.hip fit_rebuild_precond).
//
// The first preconditioner is the Ridge matrix C C^T = I + L^T L (parameters.py:895-896): the MAP Hessian
// I + L^T diag(a) L with every weight a_i = exp(f_i + V_i) set to 1 (inference.py:83-92 gives the likelihood whose second
// derivative a_i is).  At the optimum the a_i are anything but uniform -- with the nearest-neighbour likelihood in 50
// dimensions 1 % of the cells carry ~90 % of sum(a), 5 % carry 98 % (tools/precond_experiment.py) -- so the Ridge
// matrix leaves a condition number of ~50 and L-BFGS needs ~4 passes per decade.  Once the iterate is near the optimum
// (progress per iteration < 1 %) the solver therefore pauses and the Hessian THERE is estimated from an importance
// sample: cell i is kept with probability p_i = min(1, c a_i) (c such that ~12 m cells remain) and weighted a_i / p_i =
// max(a_i, 1 / c) -- an unbiased estimate whose variance does not depend on the heavy tail, where a uniform sample of
// the same size is useless (measured: 66 instead of 7 passes after the rebuild).  The choice is a deterministic hash of
// the GLOBAL cell index: every rank, every run and every sharding draw the same cells.
//
// This file only selects and gathers rows; the Gram of the gathered rows, its whitening and factorisation are the
// same routines that built the first preconditioner.
#include <algorithm>
#include <cmath>
#include <vector>

#include "mln_internal.h"
#include "precond_rebuild.h"

namespace {

constexpr int SB = 1024;   // rows per selection block

__device__ __forceinline__ double hash_unit(uint64_t i) {   // splitmix64 -> [0, 1)
  uint64_t z = i + 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  z ^= z >> 31;
  return (double)(z >> 11) * 0x1p-53;
}

__device__ __forceinline__ double block_sum256(double v, double* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// a_i = exp(f_i + V_i), clamped to a finite range; per-block partial sums and maxima
__global__ __launch_bounds__(256) void k_weights(const double* __restrict__ f, const double* __restrict__ V, int64_t n,
                                                 double* __restrict__ a, double* __restrict__ part_sum,
                                                 double* __restrict__ part_max, double cap) {
  __shared__ double red[4];
  double s = 0.0, mx = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double t = f[i] + V[i];
    t = fmin(fmax(t, -700.0), fmin(600.0, cap));     // (cap: the curvature of the solver's capped objective is e^cap above it)
    const double v = exp(t);
    a[i] = v;
    s += v;
    mx = fmax(mx, v);
  }
  s = block_sum256(s, red);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
  __shared__ double redm[4];
  if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    part_sum[blockIdx.x] = s;
    part_max[blockIdx.x] = fmax(fmax(redm[0], redm[1]), fmax(redm[2], redm[3]));
  }
}

// per-block partials of g(c) = sum_i min(1, c a_i) and of its slope g'(c) = sum of the a_i with c a_i < 1
__global__ __launch_bounds__(256) void k_expected(const double* __restrict__ a, int64_t n, double c,
                                                  double* __restrict__ part, double* __restrict__ part_slope) {
  __shared__ double red[4];
  double s = 0.0, sl = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double v = a[i], ca = c * v;
    s += fmin(1.0, ca);
    sl += (ca < 1.0) ? v : 0.0;
  }
  s = block_sum256(s, red);
  sl = block_sum256(sl, red);
  if (threadIdx.x == 0) { part[blockIdx.x] = s; part_slope[blockIdx.x] = sl; }
}

// fixed-order sums of both into out[0], out[1]
__global__ void k_fold2(const double* __restrict__ p0, const double* __restrict__ p1, int n_part, double* __restrict__ out) {
  if (blockIdx.x == 0 && threadIdx.x < 2) {
    const double* p = threadIdx.x == 0 ? p0 : p1;
    double s = 0.0;
    for (int i = 0; i < n_part; ++i) s += p[i];
    out[threadIdx.x] = s;
  }
}

// fixed-order sum (or max) of the block partials into out[0]
__global__ void k_fold(const double* __restrict__ part, int n_part, int take_max, double* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < n_part; ++i) s = take_max ? fmax(s, part[i]) : s + part[i];
    out[0] = s;
  }
}

__device__ __forceinline__ bool picked(double a, double c, int64_t gidx, uint64_t seed) {
  const double p = fmin(1.0, c * a);
  return hash_unit((uint64_t)gidx * 0x2545f4914f6cdd1dull + seed) < p;
}

// selected rows per block of SB rows
__global__ __launch_bounds__(256) void k_select_count(const double* __restrict__ a, int64_t n, double c, int64_t row0,
                                                      uint64_t seed, int* __restrict__ counts) {
  __shared__ int tot;
  if (threadIdx.x == 0) tot = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SB;
  int mine = 0;
  for (int j = threadIdx.x; j < SB; j += 256) {
    const int64_t i = base + j;
    if (i < n && picked(a[i], c, row0 + i, seed)) ++mine;
  }
  atomicAdd(&tot, mine);
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = tot;
}

// ... and their indices (ascending) and scales sqrt(w_i / w_max), w_i = max(a_i, 1 / c): one wave per block keeps order
__global__ __launch_bounds__(64) void k_select_write(const double* __restrict__ a, int64_t n, double c, int64_t row0,
                                                     uint64_t seed, const int64_t* __restrict__ offsets, double inv_wmax,
                                                     int64_t* __restrict__ idx, double* __restrict__ scale) {
  const int64_t base = (int64_t)blockIdx.x * SB;
  int64_t out = offsets[blockIdx.x];
  for (int j0 = 0; j0 < SB; j0 += 64) {
    const int64_t i = base + j0 + threadIdx.x;
    const bool on = i < n && picked(a[i], c, row0 + i, seed);
    const uint64_t mask = __ballot(on);
    if (on) {
      const int pos = __popcll(mask & ((1ull << threadIdx.x) - 1ull));
      idx[out + pos] = i;
      const double w = fmax(a[i], 1.0 / c);
      scale[out + pos] = sqrt(w * inv_wmax);
    }
    out += __popcll(mask);
  }
}

typedef double d2 __attribute__((ext_vector_type(2)));

// R[k][:] = scale[k] * A[idx[k]][:]   (row pitch ld doubles for both, ld even)
__global__ __launch_bounds__(256) void k_gather_scale(const double* __restrict__ A, int64_t ld, const int64_t* __restrict__ idx,
                                                      const double* __restrict__ scale, int64_t rows, double* __restrict__ R) {
  const int64_t k = blockIdx.x;
  if (k >= rows) return;
  const d2* __restrict__ src = reinterpret_cast<const d2*>(A + idx[k] * ld);
  d2* __restrict__ dst = reinterpret_cast<d2*>(R + k * ld);
  const double s = scale[k];
  for (int64_t p = threadIdx.x; p < ld / 2; p += 256) {
    const d2 v = src[p];
    dst[p] = (d2){v.x * s, v.y * s};
  }
}

}  // namespace

int rebuild_select_rows(mln_ctx* ctx, const double* f_dev, const double* V_dev, int64_t n, int64_t row0,
                        double target_rows_global, uint64_t seed, RebuildSelection* out, double cap) {
  out->rows = 0; out->idx = nullptr; out->scale = nullptr; out->w_max = 1.0; out->c = 0.0; out->sum_a = 0.0;
  const int nb = 256;
  double *a = nullptr, *part = nullptr, *part2 = nullptr, *scal = nullptr;
  const int64_t n1 = n > 0 ? n : 1;
  MLN_HIP(ctx, mln_dmalloc((void**)&a, sizeof(double) * (size_t)n1));
  auto fail = [&](int rc) {
    (void)hipStreamSynchronize(ctx->stream);
    if (a) (void)mln_dfree(a);
    if (part) (void)mln_dfree(part);
    if (part2) (void)mln_dfree(part2);
    if (scal) (void)mln_dfree(scal);
    return rc;
  };
  if (mln_dmalloc((void**)&part, sizeof(double) * nb) != hipSuccess || mln_dmalloc((void**)&part2, sizeof(double) * nb) != hipSuccess ||
      mln_dmalloc((void**)&scal, sizeof(double) * 16) != hipSuccess) {
    mln_set_error(ctx, "preconditioner rebuild: out of device memory");
    return fail(MLN_ERR_HIP);
  }
  hipLaunchKernelGGL(k_weights, dim3(nb), dim3(256), 0, ctx->stream, f_dev, V_dev, n, a, part, part2, cap);
  hipLaunchKernelGGL(k_fold, dim3(1), dim3(64), 0, ctx->stream, part, nb, 0, scal);        // sum a  (this rank)
  hipLaunchKernelGGL(k_fold, dim3(1), dim3(64), 0, ctx->stream, part2, nb, 1, scal + 1);   // max a  (this rank)
  int rc = comm_allreduce(ctx, scal, 1);
  if (rc != MLN_OK) return fail(rc);
  // the global maximum: every rank contributes its own (all-gather of one value per rank)
  const int nr = ctx->n_ranks > 1 ? ctx->n_ranks : 1;
  double* gathered = nullptr;
  if (mln_dmalloc((void**)&gathered, sizeof(double) * nr) != hipSuccess) return fail(MLN_ERR_HIP);
  rc = (nr > 1) ? comm_allgather(ctx, scal + 1, gathered, 1)
                : (hipMemcpyAsync(gathered, scal + 1, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream) == hipSuccess ? MLN_OK : MLN_ERR_HIP);
  std::vector<double> h(nr + 1);
  if (rc == MLN_OK && (hipMemcpyAsync(h.data(), gathered, sizeof(double) * nr, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                       hipMemcpyAsync(h.data() + nr, scal, sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                       hipStreamSynchronize(ctx->stream) != hipSuccess)) rc = MLN_ERR_HIP;
  (void)mln_dfree(gathered);
  if (rc != MLN_OK) return fail(rc);
  double a_max = 0.0;
  for (int r = 0; r < nr; ++r) a_max = std::max(a_max, h[r]);
  const double sum_a = h[nr];
  if (!(sum_a > 0.0) || !std::isfinite(sum_a)) { mln_set_error(ctx, "preconditioner rebuild: degenerate weights"); return fail(MLN_ERR_ARG); }
  // c with  g(c) = sum_i min(1, c a_i) = target.  g is concave, increasing and piecewise linear; Newton's iteration from the
  // left (c0 = target / sum a <= c*, since min(1, x) <= x) rises monotonically and never overshoots -- a tangent of a concave
  // function lies above it -- and lands exactly once no breakpoint is left between the iterate and c*: 3-4 rounds where the
  // multiplicative fixed point c *= target / g(c) took 9 (each round is a launch pair, an all-reduce and a host round trip:
  // 0.5 ms of a C2 step, 0.7-1 ms on 8 ranks).  Every rank computes the same c from the same all-reduced pair.
  double c = target_rows_global / sum_a;
  for (int it = 0; it < 14; ++it) {
    hipLaunchKernelGGL(k_expected, dim3(nb), dim3(256), 0, ctx->stream, a, n, c, part, part2);
    hipLaunchKernelGGL(k_fold2, dim3(1), dim3(64), 0, ctx->stream, part, part2, nb, scal + 2);
    rc = comm_allreduce(ctx, scal + 2, 2);
    double gs[2] = {0.0, 0.0};
    if (rc == MLN_OK && (hipMemcpyAsync(gs, scal + 2, sizeof(gs), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                         hipStreamSynchronize(ctx->stream) != hipSuccess)) rc = MLN_ERR_HIP;
    if (rc != MLN_OK) return fail(rc);
    const double cnt = gs[0], slope = gs[1];
    if (!(cnt > 0.0)) break;
    if (std::fabs(target_rows_global / cnt - 1.0) < 1e-3) break;
    if (!(slope > 0.0)) break;            // every cell saturated: g(c) = n < target cannot be raised
    const double c_new = c + (target_rows_global - cnt) / slope;
    if (!(c_new > c) || !std::isfinite(c_new)) break;
    c = c_new;
    if (c * a_max > 1e12) break;          // (almost) every cell is kept: nothing left to solve for
  }
  // weights are a_i / p_i = max(a_i, 1 / c); rows are scaled by sqrt(w_i / w_max) so that the scaled covariances stay
  // in [0, 1] (the range the integer Gram quantises) and the Gram is multiplied back by w_max
  const double w_max = std::max(a_max, 1.0 / c);
  const int64_t n_blk = (n + SB - 1) / SB;
  int* counts = nullptr;
  int64_t* offsets = nullptr;
  if (n_blk > 0) {
    if (mln_dmalloc((void**)&counts, sizeof(int) * (size_t)n_blk) != hipSuccess ||
        mln_dmalloc((void**)&offsets, sizeof(int64_t) * (size_t)n_blk) != hipSuccess) {
      if (counts) (void)mln_dfree(counts);
      return fail(MLN_ERR_HIP);
    }
    hipLaunchKernelGGL(k_select_count, dim3((unsigned)n_blk), dim3(256), 0, ctx->stream, a, n, c, row0, seed, counts);
    std::vector<int> hc((size_t)n_blk);
    std::vector<int64_t> ho((size_t)n_blk);
    if (hipMemcpyAsync(hc.data(), counts, sizeof(int) * (size_t)n_blk, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) rc = MLN_ERR_HIP;
    int64_t total = 0;
    for (int64_t b = 0; b < n_blk && rc == MLN_OK; ++b) { ho[(size_t)b] = total; total += hc[(size_t)b]; }
    if (rc == MLN_OK && total > 0) {
      if (mln_dmalloc((void**)&out->idx, sizeof(int64_t) * (size_t)total) != hipSuccess ||
          mln_dmalloc((void**)&out->scale, sizeof(double) * (size_t)total) != hipSuccess) rc = MLN_ERR_HIP;
      if (rc == MLN_OK && hipMemcpyAsync(offsets, ho.data(), sizeof(int64_t) * (size_t)n_blk, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        rc = MLN_ERR_HIP;
      if (rc == MLN_OK) {
        hipLaunchKernelGGL(k_select_write, dim3((unsigned)n_blk), dim3(64), 0, ctx->stream, a, n, c, row0, seed, offsets,
                           1.0 / w_max, out->idx, out->scale);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = MLN_ERR_HIP;
      }
    }
    (void)mln_dfree(counts);
    (void)mln_dfree(offsets);
    if (rc != MLN_OK) { rebuild_selection_free(ctx, out); mln_set_error(ctx, "preconditioner rebuild: row selection failed"); return fail(rc); }
    out->rows = total;
  }
  out->w_max = w_max; out->c = c; out->sum_a = sum_a;
  (void)fail(MLN_OK);
  return MLN_OK;
}

void rebuild_selection_free(mln_ctx* ctx, RebuildSelection* s) {
  (void)hipStreamSynchronize(ctx->stream);
  if (s->idx) (void)mln_dfree(s->idx);
  if (s->scale) (void)mln_dfree(s->scale);
  s->idx = nullptr; s->scale = nullptr; s->rows = 0;
}

int launch_gather_scale_rows(mln_ctx* ctx, const double* A, int64_t ld, const int64_t* idx, const double* scale,
                             int64_t rows, double* R) {
  if (rows <= 0) return MLN_OK;
  if (ld & 1) { mln_set_error(ctx, "gather rows: odd leading dimension"); return MLN_ERR_ARG; }
  hipLaunchKernelGGL(k_gather_scale, dim3((unsigned)rows), dim3(256), 0, ctx->stream, A, ld, idx, scale, rows, R);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}
