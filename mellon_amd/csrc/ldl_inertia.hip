// Rank diagnostic by inertia (util.test_rank, util.py:429-483: matrix_rank(L, rtol) = the number of singular values of L
// above tol * the largest = the number of eigenvalues of the m x m Gram G = L^T L above tol^2 lambda_max).
//
// tridiag.hip answers that with a Householder tridiagonalisation (4/3 m^3 flops, half of them memory-bound matrix-vector
// work: 0.25 s at m = 5000).  A COUNT needs less: by Sylvester's law of inertia the number of eigenvalues of G below x is
// the number of negative pivots of ANY factorisation G - x I = T S T^T with S = diag(+-1), and the blocked right-looking
// factorisation of linalg.hip produces one in m^3 / 3 matrix-core flops once its diagonal-block kernel accepts negative
// pivots (potrf.hip, k_potrf128<SIGNED>):
//   per 128-wide block column:  T_kk S_k T_kk^T = A_kk,   M = A_panel T_kk^-T  (= L S_k),   L = M S_k,   A22 -= M L^T.
// lambda_max comes from a plain Lanczos recurrence (the largest Ritz value converges first and is unharmed by the loss of
// orthogonality; G v is one 200 MB read), x = tol^2 lambda_max.
// No pivoting: the factorisation exists whenever no leading block is singular, and it is accurate as long as no pivot is
// tiny against x (element growth ~ 1 / |pivot|).  The smallest |pivot| is recorded; below 1e-9 x -- or on a zero / NaN
// pivot -- the caller is told to use the tridiagonal path instead (G - x I with a handful of huge and thousands of near-zero
// eigenvalues has pivots near -x or far above it; the guard is for the coincidence).
#include <cmath>
#include <cstring>
#include <vector>

#include "mln_internal.h"
#include "linalg.h"

int launch_potrf128_signed(mln_ctx* ctx, double* A, int64_t lda, int nb, double* Dinv, double* DinvS, int* info, int64_t j0,
                           int* n_neg, unsigned long long* min_piv);

namespace {

constexpr int LZ = 1024;

__device__ __forceinline__ double lz_block_sum(double v, double* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < LZ / 64; ++w) s += red[w];
  return s;
}

// start vector: positive entries with a deterministic scatter, normalised (the Gram of a covariance with non-negative
// values has a positive Perron vector, never orthogonal to this; for a general Gram orthogonality to the leading
// eigenvector would be a coincidence that rounding removes within a few steps)
__global__ __launch_bounds__(LZ) void k_lz_init(double* __restrict__ v, double* __restrict__ vprev, int64_t m) {
  __shared__ double red[LZ / 64];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < m; i += LZ) {
    const unsigned h = (unsigned)i * 2654435761u;
    const double x = 1.0 + 0.5 * ((double)(h >> 8) / 16777216.0);
    v[i] = x; vprev[i] = 0.0;
    s = fma(x, x, s);
  }
  s = lz_block_sum(s, red);
  const double inv = 1.0 / sqrt(s);
  for (int64_t i = threadIdx.x; i < m; i += LZ) v[i] *= inv;
}

// w = A v, one wave per row (rows are contiguous)
__global__ __launch_bounds__(256) void k_lz_symv(const double* __restrict__ A, int64_t ld, int64_t m,
                                                 const double* __restrict__ v, double* __restrict__ w) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  const double* a = A + row * ld;
  double acc = 0.0;
  for (int64_t j = threadIdx.x & 63; j < m; j += 64) acc = fma(a[j], v[j], acc);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) w[row] = acc;
}

// one Lanczos step after w = A v_j:  alpha_j = v_j.w;  w -= alpha_j v_j + beta_j v_{j-1};  beta_{j+1} = |w|;
// v_{j-1} <- v_j, v_j <- w / beta_{j+1}   (ab: alpha at [j], beta at [steps + j]; beta_0 = 0)
__global__ __launch_bounds__(LZ) void k_lz_step(int64_t m, double* __restrict__ v, double* __restrict__ vprev,
                                                double* __restrict__ w, double* __restrict__ ab, int j, int steps) {
  __shared__ double red[LZ / 64];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < m; i += LZ) s = fma(v[i], w[i], s);
  const double alpha = lz_block_sum(s, red);
  const double beta = ab[steps + j];
  double q = 0.0;
  for (int64_t i = threadIdx.x; i < m; i += LZ) {
    const double t = w[i] - alpha * v[i] - beta * vprev[i];
    w[i] = t;
    q = fma(t, t, q);
  }
  const double bn = sqrt(lz_block_sum(q, red));
  const double inv = (bn > 0.0) ? 1.0 / bn : 0.0;
  for (int64_t i = threadIdx.x; i < m; i += LZ) { vprev[i] = v[i]; v[i] = w[i] * inv; }
  if (threadIdx.x == 0) { ab[j] = alpha; ab[steps + j + 1] = bn; }
}

// eigenvalues of the symmetric tridiagonal (d, e) strictly below x
int64_t sturm_below(const std::vector<double>& d, const std::vector<double>& e, size_t k, double x, double tiny) {
  int64_t cnt = 0;
  double q = d[0] - x;
  if (q < 0.0) ++cnt;
  for (size_t i = 1; i < k; ++i) {
    if (std::fabs(q) < tiny) q = (q < 0.0) ? -tiny : tiny;
    q = d[i] - x - e[i - 1] * e[i - 1] / q;
    if (q < 0.0) ++cnt;
  }
  return cnt;
}

double tridiag_lambda_max(const std::vector<double>& d, const std::vector<double>& e, size_t k) {
  double lo = d[0], hi = d[0], nrm = 0.0;
  for (size_t i = 0; i < k; ++i) {
    const double r = (i > 0 ? std::fabs(e[i - 1]) : 0.0) + (i + 1 < k ? std::fabs(e[i]) : 0.0);
    lo = std::min(lo, d[i] - r); hi = std::max(hi, d[i] + r);
    nrm = std::max(nrm, std::fabs(d[i]) + r);
  }
  const double tiny = std::max(nrm, 1e-300) * 1e-300 + nrm * 2.3e-16 * 1e-3;
  double a = lo, b = hi;
  for (int it = 0; it < 200 && (b - a) > 4e-16 * std::max(std::fabs(a), std::fabs(b)) + 1e-300; ++it) {
    const double mid = 0.5 * (a + b);
    if (sturm_below(d, e, k, mid, tiny) >= (int64_t)k) b = mid; else a = mid;
  }
  return 0.5 * (a + b);
}

// |last component| of the unit eigenvector of the k x k tridiagonal (d, e) for its eigenvalue lam, by the three-term
// recurrence (rescaled as it goes).  With the NEXT off-diagonal beta_k this is the residual of the Ritz pair in the full
// matrix: |A y - lam y| = beta_k |s_k|.
double tridiag_last_component(const std::vector<double>& d, const std::vector<double>& e, size_t k, double lam) {
  if (k <= 1) return 1.0;
  double pm = 0.0, p = 1.0, sq = 1.0;                 // p_{j-1}, p_j, sum of squares so far
  for (size_t j = 0; j + 1 < k; ++j) {
    const double ej = e[j];
    if (!(std::fabs(ej) > 0.0)) return 1.0;           // (a split matrix: no bound from here)
    double pn = ((lam - d[j]) * p - (j > 0 ? e[j - 1] * pm : 0.0)) / ej;
    pm = p; p = pn;
    sq += p * p;
    if (sq > 1e200) { const double s = 1e-100; pm *= s; p *= s; sq *= s * s; }
  }
  return std::isfinite(p) && sq > 0.0 ? std::fabs(p) / std::sqrt(sq) : 1.0;
}

}  // namespace

// lambda_max of the symmetric A (m x m, full storage, untouched) by Lanczos; *converged = false when 480 steps did not
// settle the largest Ritz value to 1e-14
int dev_sym_lambda_max(mln_ctx* ctx, const double* A, int64_t m, int64_t ld, double* lambda_max, bool* converged) {
  *converged = false;
  *lambda_max = 0.0;
  if (m <= 0) { *converged = true; return MLN_OK; }
  const int chunk = 24, max_steps = 480;
  double* work = nullptr;   // v, vprev, w (m each), alpha[max_steps], beta[max_steps + 1]
  MLN_HIP(ctx, mln_dmalloc((void**)&work, sizeof(double) * (size_t)(3 * m + 2 * max_steps + 8)));
  double *v = work, *vprev = work + m, *w = work + 2 * m, *ab = work + 3 * m;
  hipError_t err = hipMemsetAsync(ab, 0, sizeof(double) * (size_t)(2 * max_steps + 8), ctx->stream);
  hipLaunchKernelGGL(k_lz_init, dim3(1), dim3(LZ), 0, ctx->stream, v, vprev, m);
  std::vector<double> hab((size_t)(2 * max_steps + 8)), d, e;
  double prev = -1.0, lmax = 0.0;
  const int64_t lim = (m < max_steps) ? m : max_steps;
  int done = 0;
  while (err == hipSuccess && done < lim) {
    const int upto = (int)std::min<int64_t>(lim, done + chunk);
    for (int j = done; j < upto; ++j) {
      hipLaunchKernelGGL(k_lz_symv, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, ctx->stream, A, ld, m, v, w);
      hipLaunchKernelGGL(k_lz_step, dim3(1), dim3(LZ), 0, ctx->stream, m, v, vprev, w, ab, j, max_steps);
    }
    done = upto;
    err = hipGetLastError();
    if (err == hipSuccess) err = hipMemcpyAsync(hab.data(), ab, sizeof(double) * hab.size(), hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
    if (err != hipSuccess) break;
    d.assign(hab.begin(), hab.begin() + done);
    e.resize((size_t)(done > 1 ? done - 1 : 0));
    bool finite = true, broke = false;
    for (int j = 0; j < done; ++j) finite = finite && std::isfinite(d[(size_t)j]);
    for (int j = 0; j + 1 < done; ++j) {
      e[(size_t)j] = hab[(size_t)(max_steps + j + 1)];
      finite = finite && std::isfinite(e[(size_t)j]);
    }
    if (!finite) { (void)mln_dfree(work); mln_set_error(ctx, "rank diagnostic: the Gram is not finite"); return MLN_ERR_NOCONV; }
    // an exhausted Krylov space (beta = 0: every later alpha is 0) ends the recurrence: the Ritz values so far are exact
    size_t k = (size_t)done;
    for (int j = 0; j < done; ++j)
      if (!(hab[(size_t)(max_steps + j + 1)] > 0.0)) { k = (size_t)j + 1; broke = true; break; }
    lmax = tridiag_lambda_max(d, e, k);
    // settled = two Ritz values 24 steps apart agree AND the Ritz pair's residual beta_k |s_k| is small: agreement alone is
    // also what stagnation on a cluster of top eigenvalues looks like, and a lambda_max that is too small lowers the
    // threshold tol^2 lambda_max the ranks are counted against
    const double beta_next = hab[(size_t)(max_steps + (int)k)];
    // (an eigenvalue of A lies within the residual of the Ritz value; the Ritz value itself converges like its square)
    const bool residual_ok = beta_next * tridiag_last_component(d, e, k, lmax) <= 1e-6 * std::fabs(lmax);
    if (broke || done >= m || (prev >= 0.0 && std::fabs(lmax - prev) <= 1e-14 * std::fabs(lmax) && residual_ok)) { *converged = true; break; }
    prev = lmax;
  }
  (void)mln_dfree(work);
  if (err != hipSuccess) return mln_hip_fail(ctx, err, "lanczos", __FILE__, __LINE__);
  *lambda_max = lmax;
  return MLN_OK;
}

// A (m x m symmetric, full storage, DESTROYED) -> the number of eigenvalues above tol2 * lambda_max.  *ok = false: the
// count could not be certified here (Lanczos not settled, a singular leading block, a pivot too small) -- the caller runs
// the tridiagonal path on a fresh copy.
int dev_sym_rank_above_ldl(mln_ctx* ctx, double* A, int64_t m, int64_t ld, double tol2, int64_t* rank, double* lambda_max, bool* ok) {
  *ok = false;
  if (m <= 0) { *rank = 0; if (lambda_max) *lambda_max = 0.0; *ok = true; return MLN_OK; }
  double lmax = 0.0;
  bool conv = false;
  MLN_TRY(dev_sym_lambda_max(ctx, A, m, ld, &lmax, &conv));
  if (lambda_max) *lambda_max = lmax;
  if (!conv) return MLN_OK;
  if (!(lmax > 0.0)) { *rank = 0; *ok = true; return MLN_OK; }
  const double thr = std::nextafter(tol2 * lmax, INFINITY);
  constexpr int CB = 128;
  double *Dinv = nullptr, *Ls = nullptr, *Ls2 = nullptr;
  int* cnt = nullptr;      // [0] negative pivots, [2..3] bits of the smallest |pivot|
  auto release = [&]() {
    (void)hipStreamSynchronize(ctx->stream);
    if (Dinv) (void)mln_dfree(Dinv);
    if (cnt) (void)mln_dfree(cnt);
    if (Ls) (void)mln_dfree(Ls);
    if (Ls2) (void)mln_dfree(Ls2);
  };
  {
    hipError_t ea = mln_dmalloc((void**)&Dinv, sizeof(double) * 2 * CB * CB);
    if (ea == hipSuccess) ea = mln_dmalloc((void**)&cnt, 16);
    if (ea == hipSuccess && m > CB) ea = mln_dmalloc((void**)&Ls, sizeof(double) * (size_t)m * (size_t)ld);
    if (ea == hipSuccess && m > CB) ea = mln_dmalloc((void**)&Ls2, sizeof(double) * (size_t)m * (size_t)ld);
    if (ea != hipSuccess) { release(); return mln_hip_fail(ctx, ea, "ldl work space", __FILE__, __LINE__); }
  }
  double* DinvS = Dinv + CB * CB;
  unsigned long long* minp = reinterpret_cast<unsigned long long*>(cnt + 2);
  const unsigned long long inf_bits = 0x7ff0000000000000ULL;
  int rc = MLN_OK;
  hipError_t e = hipMemsetAsync(Dinv, 0, sizeof(double) * 2 * CB * CB, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(cnt, 0, 8, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(minp, &inf_bits, 8, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(ctx->d_info, 0, sizeof(int), ctx->stream);
  if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "ldl setup", __FILE__, __LINE__);
  if (rc == MLN_OK) rc = launch_add_diag(ctx, A, m, ld, -thr);
  for (int64_t j0 = 0; j0 < m && rc == MLN_OK; j0 += CB) {
    const int nb = (int)((m - j0 < CB) ? (m - j0) : CB);
    rc = launch_potrf128_signed(ctx, A + j0 * ld + j0, ld, nb, Dinv, DinvS, ctx->d_info, j0, cnt, minp);
    const int64_t rem = m - j0 - nb;
    if (rc != MLN_OK || rem <= 0) break;
    GemmArgs g{};                       // M = A_panel T^-T  and  L = M S
    g.A = A + (j0 + nb) * ld + j0; g.lda = ld; g.B = Dinv; g.ldb = CB; g.C = Ls + (j0 + nb) * ld + j0; g.ldc = ld;
    g.M = rem; g.N = nb; g.K = nb; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 1;
    rc = launch_dgemm(ctx, g);
    if (rc != MLN_OK) break;
    g.B = DinvS; g.C = Ls2 + (j0 + nb) * ld + j0;
    rc = launch_dgemm(ctx, g);
    if (rc != MLN_OK) break;
    GemmArgs t{};                       // A22 -= M L^T on the lower tiles
    t.A = Ls + (j0 + nb) * ld + j0; t.lda = ld; t.B = Ls2 + (j0 + nb) * ld + j0; t.ldb = ld;
    t.C = A + (j0 + nb) * ld + (j0 + nb); t.ldc = ld;
    t.M = rem; t.N = rem; t.K = nb; t.alpha = -1.0; t.beta = 1.0; t.ta = 0; t.tb = 1; t.lower_only = 1;
    rc = launch_dgemm(ctx, t);
  }
  int hcnt[4] = {0, 0, 0, 0};
  int info = 0;
  if (rc == MLN_OK) {
    e = hipMemcpyAsync(hcnt, cnt, 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "ldl sync", __FILE__, __LINE__);
  } else {
    (void)hipStreamSynchronize(ctx->stream);
  }
  release();
  if (rc != MLN_OK) return rc;
  unsigned long long bits = 0;
  std::memcpy(&bits, hcnt + 2, 8);
  double minpiv = 0.0;
  std::memcpy(&minpiv, &bits, 8);
  if (info != 0 || !(minpiv > 1e-9 * thr)) return MLN_OK;      // not certified here: *ok stays false
  *rank = m - (int64_t)hcnt[0];
  *ok = true;
  return MLN_OK;
}
