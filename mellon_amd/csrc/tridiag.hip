// Rank diagnostic without an eigendecomposition (util.test_rank, util.py:429-483: matrix_rank(L, rtol) = number of
// singular values above tol * the largest).  The singular values of L are the square roots of the eigenvalues of the
// m x m Gram G = L^T L; a COUNT of eigenvalues above a threshold needs no eigenvectors and no eigenvalues either:
// Householder tridiagonalisation (backward stable, 4/3 m^3 flops, two passes over the shrinking trailing matrix per
// step: memory-bound, ~0.7 TB of traffic at m = 5000) and Sturm-sequence counts on the tridiagonal matrix -- Sylvester's
// law of inertia applied to T - x I -- give lambda_max by bisection and then the count above tol^2 lambda_max exactly.
// 0.4 s at m = 5000 where the block-Jacobi eigensolver (eigh.hip, needed for the Nystroem factors) takes ~3 s.
#include "mln_internal.h"

#include <cmath>
#include <vector>

namespace {

constexpr int HT = 1024;

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ double block_sum_d(double v, double* red) {   // HT threads
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < HT / 64; ++w) s += red[w];
  return s;
}

// Step k, part 1 (LAPACK dlarfg on x = A[k, k+1:], the mirror image of column k below the diagonal): v (v_0 = 1), tau,
// d[k] = A[k][k], e[k] = beta.
__global__ __launch_bounds__(HT) void k_house(const double* __restrict__ A, int64_t ld, int64_t n, int64_t k,
                                              double* __restrict__ v, double* __restrict__ tau, double* __restrict__ d,
                                              double* __restrict__ e) {
  __shared__ double red[HT / 64];
  const int64_t s = n - k - 1;
  const double* x = A + k * ld + k + 1;
  // dlarfg's safeguard: the sum of squares is taken of x / max|x|, so that a Gram with entries near the ends of the
  // double range neither overflows nor underflows into a NaN (or a zero) beta
  double mx = 0.0;
  for (int64_t i = threadIdx.x; i < s; i += HT) mx = fmax(mx, fabs(x[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = 0.0;
#pragma unroll
  for (int w = 0; w < HT / 64; ++w) mx = fmax(mx, red[w]);
  const double inv = (mx > 0.0 && isfinite(mx)) ? 1.0 / mx : 0.0;
  double sig = 0.0;
  for (int64_t i = 1 + threadIdx.x; i < s; i += HT) { const double xi = x[i] * inv; sig = fma(xi, xi, sig); }
  sig = block_sum_d(sig, red);
  const double alpha = x[0];
  double beta = alpha, t = 0.0, scale = 0.0;
  if (sig > 0.0) {
    const double as = alpha * inv;
    beta = -copysign(mx * sqrt(fma(as, as, sig)), alpha);
    t = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
  }
  for (int64_t i = threadIdx.x; i < s; i += HT) v[i] = (i == 0) ? 1.0 : x[i] * scale;
  if (threadIdx.x == 0) { *tau = t; d[k] = A[k * ld + k]; e[k] = beta; }
}

// part 2: p = A22 v over the trailing s x s block (one wave per row; rows are contiguous)
__global__ __launch_bounds__(256) void k_symv_rows(const double* __restrict__ A, int64_t ld, int64_t n, int64_t k,
                                                   const double* __restrict__ v, double* __restrict__ p) {
  const int64_t s = n - k - 1;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= s) return;
  const double* a = A + (k + 1 + row) * ld + k + 1;
  double acc = 0.0;
  for (int64_t j = threadIdx.x & 63; j < s; j += 64) acc = fma(a[j], v[j], acc);
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) p[row] = acc;
}

// part 3 (LAPACK dsytd2): p <- tau p ; w = p - (tau / 2) (p . v) v
__global__ __launch_bounds__(HT) void k_house_w(int64_t s, const double* __restrict__ v, const double* __restrict__ p,
                                                const double* __restrict__ tau, double* __restrict__ w) {
  __shared__ double red[HT / 64];
  const double t = *tau;
  double dot = 0.0;
  for (int64_t i = threadIdx.x; i < s; i += HT) dot = fma(t * p[i], v[i], dot);
  dot = block_sum_d(dot, red);
  const double a = -0.5 * t * dot;
  for (int64_t i = threadIdx.x; i < s; i += HT) w[i] = fma(a, v[i], t * p[i]);
}

// part 4: A22 -= v w^T + w v^T (the full square: row k + 1 of the result is the next step's x)
__global__ __launch_bounds__(256) void k_rank2_rows(double* __restrict__ A, int64_t ld, int64_t n, int64_t k,
                                                    const double* __restrict__ v, const double* __restrict__ w) {
  const int64_t s = n - k - 1;
  const int64_t row = blockIdx.x;
  const double vi = v[row], wi = w[row];
  double* a = A + (k + 1 + row) * ld + k + 1;
  for (int64_t j = threadIdx.x; j < s; j += 256) a[j] -= fma(vi, w[j], wi * v[j]);
}

// eigenvalues of the symmetric tridiagonal (d, e) strictly below x (Sturm sequence of the LDL^T pivots of T - x I)
int64_t sturm_count(const std::vector<double>& d, const std::vector<double>& e, double x, double tiny) {
  int64_t cnt = 0;
  double q = d[0] - x;
  if (q < 0.0) ++cnt;
  for (size_t i = 1; i < d.size(); ++i) {
    if (std::fabs(q) < tiny) q = (q < 0.0) ? -tiny : tiny;
    q = d[i] - x - e[i - 1] * e[i - 1] / q;
    if (q < 0.0) ++cnt;
  }
  return cnt;
}

}  // namespace

// A (m x m symmetric, full storage, DESTROYED) -> number of eigenvalues above tol2 * lambda_max; lambda_max returned too
int dev_sym_rank_above(mln_ctx* ctx, double* A, int64_t m, int64_t ld, double tol2, int64_t* rank, double* lambda_max) {
  if (m <= 0) { *rank = 0; if (lambda_max) *lambda_max = 0.0; return MLN_OK; }
  double* work = nullptr;   // v, p, w (m each), d, e (m each), tau
  MLN_HIP(ctx, mln_dmalloc((void**)&work, sizeof(double) * (size_t)(5 * m + 8)));
  double *v = work, *p = work + m, *w = work + 2 * m, *dd = work + 3 * m, *de = work + 4 * m, *tau = work + 5 * m;
  for (int64_t k = 0; k + 2 < m; ++k) {
    const int64_t s = m - k - 1;
    hipLaunchKernelGGL(k_house, dim3(1), dim3(HT), 0, ctx->stream, A, ld, m, k, v, tau, dd, de);
    hipLaunchKernelGGL(k_symv_rows, dim3((unsigned)((s + 3) / 4)), dim3(256), 0, ctx->stream, A, ld, m, k, v, p);
    hipLaunchKernelGGL(k_house_w, dim3(1), dim3(HT), 0, ctx->stream, s, v, p, tau, w);
    hipLaunchKernelGGL(k_rank2_rows, dim3((unsigned)s), dim3(256), 0, ctx->stream, A, ld, m, k, v, w);
  }
  hipError_t err = hipGetLastError();
  // (the last 2 x 2 block -- or the 1 x 1 matrix -- comes straight from A)
  std::vector<double> d((size_t)m), e((size_t)(m > 1 ? m - 1 : 0));
  if (err == hipSuccess) err = hipMemcpyAsync(d.data(), dd, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream);
  if (err == hipSuccess && m > 1) err = hipMemcpyAsync(e.data(), de, sizeof(double) * (m - 1), hipMemcpyDeviceToHost, ctx->stream);
  double tail[3] = {0.0, 0.0, 0.0};   // A[m-2][m-2], A[m-2][m-1], A[m-1][m-1]
  if (err == hipSuccess && m >= 2) {
    err = hipMemcpyAsync(&tail[0], A + (m - 2) * ld + (m - 2), sizeof(double) * 2, hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess) err = hipMemcpyAsync(&tail[2], A + (m - 1) * ld + (m - 1), sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  } else if (err == hipSuccess) {
    err = hipMemcpyAsync(&tail[2], A, sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  }
  if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
  (void)mln_dfree(work);
  if (err != hipSuccess) return mln_hip_fail(ctx, err, "tridiagonalisation", __FILE__, __LINE__);
  if (m >= 2) { d[m - 2] = tail[0]; e[m - 2] = tail[1]; }
  d[m - 1] = tail[2];
  for (int64_t i = 0; i < m; ++i)
    if (!std::isfinite(d[i]) || (i + 1 < m && !std::isfinite(e[i]))) {
      mln_set_error(ctx, "rank diagnostic: the tridiagonalised Gram is not finite (NaN / inf in L, or a Gram beyond the double range)");
      return MLN_ERR_NOCONV;
    }
  // Gershgorin bounds, then bisection on the Sturm count for lambda_max
  double lo = d[0], hi = d[0], nrm = 0.0;
  for (int64_t i = 0; i < m; ++i) {
    const double r = (i > 0 ? std::fabs(e[i - 1]) : 0.0) + (i + 1 < m ? std::fabs(e[i]) : 0.0);
    lo = std::min(lo, d[i] - r); hi = std::max(hi, d[i] + r);
    nrm = std::max(nrm, std::fabs(d[i]) + r);
  }
  const double tiny = std::max(nrm, 1e-300) * 1e-300 + nrm * 2.3e-16 * 1e-3;
  double a = lo, b = hi;
  for (int it = 0; it < 200 && (b - a) > 4e-16 * std::max(std::fabs(a), std::fabs(b)) + 1e-300; ++it) {
    const double mid = 0.5 * (a + b);
    if (sturm_count(d, e, mid, tiny) >= m) b = mid; else a = mid;   // all m eigenvalues below mid: mid is above lambda_max
  }
  const double lmax = 0.5 * (a + b);
  if (lambda_max) *lambda_max = lmax;
  const double thr = tol2 * lmax;
  *rank = (lmax > 0.0) ? (m - sturm_count(d, e, std::nextafter(thr, INFINITY), tiny)) : 0;   // eigenvalues > thr
  return MLN_OK;
}
