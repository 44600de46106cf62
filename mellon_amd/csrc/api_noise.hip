// C ABI, part 5: the FunctionEstimator's noisy conditionals, leverage and variance weights (see api_internal.h).
#include "api_internal.h"

// ---- FunctionEstimator sparse solve ----------------------------------------------------------------
// C[i][j] *= (row ? row[i] : 1) * (col ? col[j] : 1)
__global__ void k_scale_rows_cols(double* __restrict__ A, int64_t ld, int64_t rows, int64_t cols,
                                  const double* __restrict__ row, const double* __restrict__ col) {
  const int64_t i = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (i >= rows) return;
  const double ri = row ? row[i] : 1.0;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cols; j += (int64_t)gridDim.x * blockDim.x)
    A[i * ld + j] *= ri * (col ? col[j] : 1.0);
}

// T[k][j] /= lam[k] * inv_s2[j] + 1  -- the resolvent (G / s_j^2 + I)^-1 in the eigenbasis of G
__global__ void k_resolvent_scale(double* __restrict__ T, int64_t ld, int64_t rows, int64_t cols,
                                  const double* __restrict__ lam, const double* __restrict__ inv_s2) {
  const int64_t k = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (k >= rows) return;
  const double l = lam[k] > 0.0 ? lam[k] : 0.0;       // A A^T is positive semi-definite; rounding may say -1e-13
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cols; j += (int64_t)gridDim.x * blockDim.x)
    T[k * ld + j] /= l * inv_s2[j] + 1.0;
}

int launch_scale_rows_cols(mln_ctx* ctx, double* A, int64_t ld, int64_t rows, int64_t cols, const double* row,
                                  const double* col) {
  if (rows <= 0 || cols <= 0) return MLN_OK;
  int64_t bx = (cols + 255) / 256;
  if (bx > 64) bx = 64;
  const int64_t by = rows < 65535 ? rows : 65535, bz = (rows + 65534) / 65535;
  hipLaunchKernelGGL(k_scale_rows_cols, dim3((unsigned)bx, (unsigned)by, (unsigned)bz), dim3(256), 0, ctx->stream, A, ld,
                     rows, cols, row, col);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

// Noise models of the landmark conditional (conditional.py:140-159, 529-545):
//   MLN_SIGMA_SCALAR      sigma[1]        r / s^2, A / s^2
//   MLN_SIGMA_PER_OUTPUT  sigma[p]        one scalar solve per output column ("per-gene", the vmap of :529-545);
//                                         A A^T and A r are formed once, columns with equal sigma share L_B
//   MLN_SIGMA_PER_CELL    sigma[n_local]  element-wise std of the cells: A diag(1/s^2) A^T and A (r / s^2)
static constexpr int SPECTRAL_MIN_LEVELS = 32;   // runs of equal sigma above which the per-output solve goes spectral

int sparse_solve_impl(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local,
                             int32_t d, const double* xu, int64_t m, const double* y, int64_t p, double mu,
                             const double* sigmas, int32_t kind, double jitter, double* W, double* Lp_out,
                             double* Cs_out) {
  if (!ctx || !xu || !W || !sigmas || (n_local > 0 && (!x || !y))) return MLN_ERR_ARG;
  if (p < 1 || m < 1) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  if (kind < MLN_SIGMA_SCALAR || kind > MLN_SIGMA_PER_CELL) { mln_set_error(ctx, "unknown sigma kind"); return MLN_ERR_ARG; }
  if (kind != MLN_SIGMA_SCALAR && (Lp_out || Cs_out)) {
    mln_set_error(ctx, "the L_B factor is only defined for one scalar sigma (conditional.py:574-577)");
    return MLN_ERR_ARG;
  }
  const int64_t n_sig = (kind == MLN_SIGMA_SCALAR) ? 1 : (kind == MLN_SIGMA_PER_OUTPUT ? p : n_local);
  for (int64_t i = 0; i < n_sig; ++i)
    if (!(sigmas[i] > 0.0)) {
      mln_set_error(ctx, "sigma must be > 0 for the sparse solve (conditional.py:157-159 divides by sigma^2)");
      return MLN_ERR_ARG;
    }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  // A^T = cov(x, xu) Lp^-T is exactly the factor L of the density path   conditional.py:516-522
  mln_fit* f = nullptr;
  MLN_TRY(mln_fit_prepare(ctx, cov, x, n_local, d, xu, m, jitter, nullptr, 0, &f));
  const int64_t ldg = pad16(m), n = n_local;
  // groups of adjacent output columns with one noise level
  std::vector<int64_t> g_begin;
  std::vector<double> g_s2;
  if (kind == MLN_SIGMA_PER_OUTPUT) {
    for (int64_t j = 0; j < p; ++j)
      if (j == 0 || sigmas[j] != sigmas[j - 1]) { g_begin.push_back(j); g_s2.push_back(sigmas[j] * sigmas[j]); }
  } else {
    g_begin.push_back(0);
    g_s2.push_back(kind == MLN_SIGMA_SCALAR ? sigmas[0] * sigmas[0] : 1.0);
  }
  g_begin.push_back(p);
  const size_t n_groups = g_s2.size();
  double *G = nullptr, *G0 = nullptr, *R = nullptr, *C = nullptr, *parts = nullptr, *d_scale = nullptr, *d_col = nullptr;
  TriInv tb;
  int rc = MLN_OK;
  auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == MLN_OK) rc = mln_hip_fail(ctx, e, "sparse_solve", __FILE__, __LINE__); };
  chk(mln_dmalloc((void**)&G, sizeof(double) * (size_t)m * ldg));
  if (n_groups > 1) chk(mln_dmalloc((void**)&G0, sizeof(double) * (size_t)m * ldg));
  chk(mln_dmalloc((void**)&C, sizeof(double) * (size_t)m * p));
  DevIn dy;
  if (rc == MLN_OK) rc = dy.init(ctx, y, (size_t)n * p);
  // r = y - mu
  if (rc == MLN_OK && n > 0) {
    chk(mln_dmalloc((void**)&R, sizeof(double) * (size_t)n * p));
    chk(hipMemcpyAsync(R, dy.dev, sizeof(double) * (size_t)n * p, hipMemcpyDeviceToDevice, ctx->stream));
    if (rc == MLN_OK && mu != 0.0) {
      std::vector<double> ones((size_t)n * p, 1.0);
      double* d1 = nullptr;
      chk(mln_dmalloc((void**)&d1, sizeof(double) * ones.size()));
      chk(hipMemcpyAsync(d1, ones.data(), sizeof(double) * ones.size(), hipMemcpyHostToDevice, ctx->stream));
      if (rc == MLN_OK) rc = launch_axpby(ctx, (int64_t)ones.size(), -mu, d1, 1.0, R);
      (void)hipStreamSynchronize(ctx->stream);
      if (d1) (void)mln_dfree(d1);
    }
  }
  // per-cell noise: rows of A^T and of r divided by sigma_i, after which the solve is the sigma = 1 one
  if (rc == MLN_OK && kind == MLN_SIGMA_PER_CELL && n > 0) {
    std::vector<double> inv((size_t)n);
    for (int64_t i = 0; i < n; ++i) inv[(size_t)i] = 1.0 / sigmas[i];
    chk(mln_dmalloc((void**)&d_scale, sizeof(double) * (size_t)n));
    chk(hipMemcpyAsync(d_scale, inv.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    if (rc == MLN_OK) rc = launch_scale_rows_cols(ctx, f->L, f->ldl, n, m, d_scale, nullptr);
    if (rc == MLN_OK) rc = launch_scale_rows_cols(ctx, R, p, n, p, d_scale, nullptr);
    (void)hipStreamSynchronize(ctx->stream);
  }
  // A A^T (all-reduced), kept when several noise levels need it
  if (rc == MLN_OK) rc = fit_gram(f, n_groups > 1 ? G0 : G, ldg, 1);
  // C = A r = L^T r   (m x p), split over cells; the 1 / sigma^2 is applied per group below
  if (rc == MLN_OK) {
    int split = (int)(n / 8192);
    if (split < 1) split = 1;
    if (split > 16) split = 16;
    const size_t stride = (size_t)m * p;
    if (split > 1) chk(mln_dmalloc((void**)&parts, sizeof(double) * stride * split));
    if (rc == MLN_OK) chk(hipMemsetAsync(split > 1 ? parts : C, 0, sizeof(double) * stride * (split > 1 ? split : 1), ctx->stream));
    GemmArgs g{};
    g.A = f->L; g.lda = f->ldl; g.B = R; g.ldb = p; g.C = (split > 1) ? parts : C; g.ldc = p;
    g.M = m; g.N = p; g.K = n; g.alpha = (n_groups == 1) ? 1.0 / g_s2[0] : 1.0; g.beta = 0.0; g.ta = 1; g.tb = 0;
    g.split_k = split; g.c_split_stride = (int64_t)stride;
    if (rc == MLN_OK && n > 0) rc = launch_dgemm(ctx, g);
    if (rc == MLN_OK && split > 1) rc = launch_sum_partials(ctx, parts, split, (int64_t)stride, C, (int64_t)stride, 0.0);
    if (rc == MLN_OK) rc = dev_allreduce(ctx, C, (int64_t)stride);
    if (rc == MLN_OK && n_groups > 1) {
      std::vector<double> inv((size_t)p);
      for (int64_t j = 0; j < p; ++j) inv[(size_t)j] = 1.0 / (sigmas[j] * sigmas[j]);
      chk(mln_dmalloc((void**)&d_col, sizeof(double) * (size_t)p));
      chk(hipMemcpyAsync(d_col, inv.data(), sizeof(double) * (size_t)p, hipMemcpyHostToDevice, ctx->stream));
      if (rc == MLN_OK) rc = launch_scale_rows_cols(ctx, C, p, m, p, nullptr, d_col);
      (void)hipStreamSynchronize(ctx->stream);
    }
  }
  if (rc == MLN_OK && n_groups > (size_t)SPECTRAL_MIN_LEVELS) {
    // Many noise levels: one eigendecomposition A A^T = U diag(lam) U^T serves them all,
    //   (A A^T / s^2 + I)^-1 c = U diag(1 / (lam / s^2 + 1)) U^T c,
    // O(m^3 + m^2 p) instead of one m^3/3 Cholesky per level.  The matrix inverted has eigenvalues >= 1, so the
    // spectral form is as well conditioned as the factorisation it replaces.
    double *V = nullptr, *T = nullptr, *d_lam = nullptr;
    std::vector<double> lam((size_t)m);
    int sweeps = 0;
    chk(mln_dmalloc((void**)&V, sizeof(double) * (size_t)m * ldg));
    chk(mln_dmalloc((void**)&T, sizeof(double) * (size_t)m * p));
    chk(mln_dmalloc((void**)&d_lam, sizeof(double) * (size_t)m));
    if (rc == MLN_OK) rc = dev_eigh(ctx, G0, m, ldg, lam.data(), V, ldg, &sweeps);     // row k of V = eigenvector k
    if (rc == MLN_OK) chk(hipMemcpyAsync(d_lam, lam.data(), sizeof(double) * (size_t)m, hipMemcpyHostToDevice, ctx->stream));
    GemmArgs g{};
    g.A = V; g.lda = ldg; g.B = C; g.ldb = p; g.C = T; g.ldc = p;
    g.M = m; g.N = p; g.K = m; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 0; g.split_k = 1;
    if (rc == MLN_OK) rc = launch_dgemm(ctx, g);                                      // U^T c
    if (rc == MLN_OK) {
      int64_t bx = (p + 255) / 256;
      if (bx > 64) bx = 64;
      hipLaunchKernelGGL(k_resolvent_scale, dim3((unsigned)bx, (unsigned)(m < 65535 ? m : 65535), (unsigned)((m + 65534) / 65535)),
                         dim3(256), 0, ctx->stream, T, p, m, p, d_lam, d_col);
      chk(hipGetLastError());
    }
    g.A = V; g.B = T; g.C = C; g.ta = 1;
    if (rc == MLN_OK) rc = launch_dgemm(ctx, g);                                      // U (.)
    (void)hipStreamSynchronize(ctx->stream);
    void* tmp[] = {V, T, d_lam};
    for (void* q : tmp) if (q) (void)mln_dfree(q);
  } else {
    for (size_t gi = 0; gi < n_groups && rc == MLN_OK; ++gi) {
      const int64_t c0 = g_begin[gi], nc = g_begin[gi + 1] - c0;
      // LBB = A A^T / sigma^2 + I                                      conditional.py:62 (stabilize(.., 1))
      rc = launch_axpby(ctx, m * ldg, 1.0 / g_s2[gi], n_groups > 1 ? G0 : G, 0.0, G);
      if (rc == MLN_OK) rc = launch_add_diag(ctx, G, m, ldg, 1.0);
      if (rc == MLN_OK) rc = dev_cholesky_lower(ctx, G, m, ldg);
      // L_B^-T L_B^-1 C                                                conditional.py:64-65
      if (rc == MLN_OK) rc = triinv_build(ctx, G, m, ldg, true, true, &tb);
      if (rc == MLN_OK) rc = triinv_solve_left(ctx, tb, C + c0, nc, p);
      if (rc == MLN_OK) rc = triinv_solve_left_T(ctx, tb, C + c0, nc, p);
      if (gi + 1 < n_groups) { (void)hipStreamSynchronize(ctx->stream); triinv_free(&tb); }
    }
  }
  // weights = Lp^-T (.)
  if (rc == MLN_OK) rc = triinv_solve_left_T(ctx, f->tri, C, p, p);
  if (rc == MLN_OK) chk(hipMemcpyAsync(W, C, sizeof(double) * (size_t)m * p, hipMemcpyDefault, ctx->stream));
  // with_uncertainty state of the noisy landmark conditional: L = Lp and Cs = Lp L_B   conditional.py:571-577
  if (rc == MLN_OK && Lp_out) {
    DevOut o;
    rc = o.init(ctx, Lp_out, (size_t)m * m);
    if (rc == MLN_OK) rc = launch_copy_block(ctx, f->Lp, f->ldp, o.dev, m, m, m);
    if (rc == MLN_OK) rc = o.commit();
  }
  if (rc == MLN_OK && Cs_out) {
    double* cs = nullptr;
    chk(mln_dmalloc((void**)&cs, sizeof(double) * (size_t)m * ldg));
    if (rc == MLN_OK) chk(hipMemsetAsync(cs, 0, sizeof(double) * (size_t)m * ldg, ctx->stream));
    GemmArgs g{};
    g.A = f->Lp; g.lda = f->ldp; g.B = G; g.ldb = ldg; g.C = cs; g.ldc = ldg;
    g.M = m; g.N = m; g.K = m; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 0; g.split_k = 1;
    if (rc == MLN_OK) rc = launch_dgemm(ctx, g);
    DevOut o;
    if (rc == MLN_OK) rc = o.init(ctx, Cs_out, (size_t)m * m);
    if (rc == MLN_OK) rc = launch_copy_block(ctx, cs, ldg, o.dev, m, m, m);
    if (rc == MLN_OK) rc = o.commit();
    (void)hipStreamSynchronize(ctx->stream);
    if (cs) (void)mln_dfree(cs);
  }
  (void)hipStreamSynchronize(ctx->stream);
  triinv_free(&tb);
  void* ptrs[] = {G, G0, R, C, parts, d_scale, d_col};
  for (void* q : ptrs) if (q) (void)mln_dfree(q);
  fit_free(f);
  return rc;
}

extern "C" int mln_sparse_solve(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local,
                                int32_t d, const double* xu, int64_t m, const double* y, int64_t p, double mu,
                                double sigma, double jitter, double* W) {
  return sparse_solve_impl(ctx, cov, x, n_local, d, xu, m, y, p, mu, &sigma, MLN_SIGMA_SCALAR, jitter, W, nullptr, nullptr);
}

extern "C" int mln_sparse_solve_factors(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local,
                                        int32_t d, const double* xu, int64_t m, const double* y, int64_t p,
                                        double mu, double sigma, double jitter, double* W, double* Lp_out,
                                        double* Cs_out) {
  return sparse_solve_impl(ctx, cov, x, n_local, d, xu, m, y, p, mu, &sigma, MLN_SIGMA_SCALAR, jitter, W, Lp_out, Cs_out);
}

// Z <- Z o Z
__global__ void k_square_inplace(double* __restrict__ Z, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    Z[i] *= Z[i];
}

// D[k][j] = 1 / (s2[j] + theta[k])
__global__ void k_resolvent_table(double* __restrict__ D, int64_t ld, int64_t rows, int64_t cols,
                                  const double* __restrict__ theta, const double* __restrict__ s2) {
  const int64_t k = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (k >= rows) return;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < cols; j += (int64_t)gridDim.x * blockDim.x)
    D[k * ld + j] = 1.0 / (s2[j] + theta[k]);
}

// Leverage of the landmark conditional for p noise levels from one eigendecomposition.
//   h_ij = b_i^T M_j^-1 b_i,  M_j = s_j^2 K_uu + B^T B + jitter I,  B = cov(x, xu)      conditional.py:660-685
// With K_uu = Lk Lk^T and l_i = Lk^-1 b_i (the rows of the low-rank factor L = B Lk^-T):
//   M_j = Lk (s_j^2 I + N) Lk^T,  N = L^T L + jitter Lk^-1 Lk^-T = V diag(theta) V^T
//   h_ij = sum_k (V^T l_i)_k^2 / (s_j^2 + theta_k)
// i.e. one TRSM, one Gram, one m x m eigensolve and two GEMMs for all levels, instead of one m x m Cholesky and one
// n m^2 solve per level.  L is formed by the triangular solve (not from B^T B), so N is accurate to eps |N|.
extern "C" int mln_landmark_leverage(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local,
                                     int32_t d, const double* xu, int64_t m, const double* Lk, const double* sigma,
                                     int64_t p, double jitter, double* out) {
  if (!ctx || !xu || !Lk || !sigma || (n_local > 0 && (!x || !out))) return MLN_ERR_ARG;
  if (p < 1 || m < 1 || n_local < 0) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  mln_fit* f = nullptr;
  MLN_TRY(mln_fit_prepare(ctx, cov, x, n_local, d, xu, m, jitter, Lk, 0, &f));
  const int64_t ldg = pad16(m), n = n_local;
  double *N = nullptr, *Li = nullptr, *J = nullptr, *V = nullptr, *D = nullptr, *Z = nullptr, *d_theta = nullptr, *d_s2 = nullptr;
  int rc = MLN_OK;
  auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == MLN_OK) rc = mln_hip_fail(ctx, e, "landmark_leverage", __FILE__, __LINE__); };
  const size_t mm = sizeof(double) * (size_t)m * ldg;
  chk(mln_dmalloc((void**)&N, mm));
  chk(mln_dmalloc((void**)&Li, mm));
  chk(mln_dmalloc((void**)&J, mm));
  chk(mln_dmalloc((void**)&V, mm));
  chk(mln_dmalloc((void**)&D, sizeof(double) * (size_t)m * p));
  chk(mln_dmalloc((void**)&d_theta, sizeof(double) * (size_t)m));
  chk(mln_dmalloc((void**)&d_s2, sizeof(double) * (size_t)p));
  DevOut o;
  if (rc == MLN_OK && n > 0) rc = o.init(ctx, out, (size_t)n * p);
  // N = L^T L (all cells, all ranks) + jitter Lk^-1 Lk^-T
  if (rc == MLN_OK) rc = fit_gram(f, N, ldg, 1);
  if (rc == MLN_OK) chk(hipMemsetAsync(Li, 0, mm, ctx->stream));
  if (rc == MLN_OK) rc = launch_add_diag(ctx, Li, m, ldg, 1.0);
  if (rc == MLN_OK) rc = triinv_solve_left(ctx, f->tri, Li, m, ldg);                  // Lk^-1
  if (rc == MLN_OK) chk(hipMemsetAsync(J, 0, mm, ctx->stream));
  {
    GemmArgs g{};
    g.A = Li; g.lda = ldg; g.B = Li; g.ldb = ldg; g.C = J; g.ldc = ldg;
    g.M = m; g.N = m; g.K = m; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 1; g.split_k = 1;
    if (rc == MLN_OK) rc = launch_dgemm(ctx, g);
  }
  if (rc == MLN_OK) rc = launch_axpby(ctx, m * ldg, jitter, J, 1.0, N);
  std::vector<double> theta((size_t)m), s2((size_t)p);
  int sweeps = 0;
  if (rc == MLN_OK) rc = dev_eigh(ctx, N, m, ldg, theta.data(), V, ldg, &sweeps);     // row k of V = eigenvector k
  if (rc == MLN_OK) {
    double smin = sigma[0] * sigma[0];
    for (int64_t j = 0; j < p; ++j) { s2[(size_t)j] = sigma[j] * sigma[j]; if (s2[(size_t)j] < smin) smin = s2[(size_t)j]; }
    if (!(smin + theta[0] > 0.0)) {
      mln_set_error(ctx, "sigma^2 K_uu + B^T B + jitter I is not positive definite");
      rc = MLN_ERR_NOT_PD;
    }
  }
  if (rc == MLN_OK) {
    chk(hipMemcpyAsync(d_theta, theta.data(), sizeof(double) * (size_t)m, hipMemcpyHostToDevice, ctx->stream));
    chk(hipMemcpyAsync(d_s2, s2.data(), sizeof(double) * (size_t)p, hipMemcpyHostToDevice, ctx->stream));
    int64_t bx = (p + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_resolvent_table, dim3((unsigned)bx, (unsigned)(m < 65535 ? m : 65535), (unsigned)((m + 65534) / 65535)),
                       dim3(256), 0, ctx->stream, D, p, m, p, d_theta, d_s2);
    chk(hipGetLastError());
  }
  // per chunk of cells: Z = L V^T (coordinates of l_i in the eigenbasis), squared, times the resolvent table
  const int64_t chunk = (n < 32768) ? (n > 0 ? n : 1) : 32768;
  if (rc == MLN_OK) chk(mln_dmalloc((void**)&Z, sizeof(double) * (size_t)chunk * ldg));
  for (int64_t r0 = 0; r0 < n && rc == MLN_OK; r0 += chunk) {
    const int64_t rows = (n - r0 < chunk) ? n - r0 : chunk;
    GemmArgs g{};
    g.A = f->L + r0 * f->ldl; g.lda = f->ldl; g.B = V; g.ldb = ldg; g.C = Z; g.ldc = ldg;
    g.M = rows; g.N = m; g.K = m; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 1; g.split_k = 1;
    rc = launch_dgemm(ctx, g);
    if (rc == MLN_OK) {
      const int64_t count = rows * ldg;
      int64_t nb = (count + 255) / 256;
      if (nb > 16384) nb = 16384;
      hipLaunchKernelGGL(k_square_inplace, dim3((unsigned)nb), dim3(256), 0, ctx->stream, Z, count);
      chk(hipGetLastError());
    }
    GemmArgs h{};
    h.A = Z; h.lda = ldg; h.B = D; h.ldb = p; h.C = o.dev + r0 * p; h.ldc = p;
    h.M = rows; h.N = p; h.K = m; h.alpha = 1.0; h.beta = 0.0; h.ta = 0; h.tb = 0; h.split_k = 1;
    if (rc == MLN_OK) rc = launch_dgemm(ctx, h);
  }
  if (rc == MLN_OK && n > 0) rc = o.commit();
  (void)hipStreamSynchronize(ctx->stream);
  void* ptrs[] = {N, Li, J, V, D, Z, d_theta, d_s2};
  for (void* q : ptrs) if (q) (void)mln_dfree(q);
  fit_free(f);
  return rc;
}

__global__ void k_mul_inplace(double* __restrict__ T, const double* __restrict__ D, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    T[i] *= D[i];
}

__global__ void k_shift(double* __restrict__ R, const double* __restrict__ y, double mu, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    R[i] = y[i] - mu;
}

// H[i][j] <- 1 - sig2[j] * H[i][j]
__global__ void k_leverage_finish(double* __restrict__ H, int64_t count, int64_t p, const double* __restrict__ sig2) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    H[i] = 1.0 - sig2[i % p] * H[i];
}

// out = (y - mu - KW)^2 / (1 - h)^2                                 conditional.py:330-333
__global__ void k_hc3(double* __restrict__ out, const double* __restrict__ y, const double* __restrict__ kw, double mu,
                      const double* __restrict__ h, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const double r = y[i] - (mu + kw[i]), q = 1.0 - h[i];
    out[i] = (r * r) / (q * q);
  }
}

unsigned grid_1d(int64_t count) {
  int64_t nb = (count + 255) / 256;
  return (unsigned)(nb > 16384 ? 16384 : (nb < 1 ? 1 : nb));
}

// Full GP conditioned on p outputs, each with its own noise level (conditional.py:239-251), from one
// eigendecomposition K = U diag(lam) U^T:  (K + s_j I)^-1 = U diag(1 / (lam + s_j)) U^T,  s_j = sigma_j^2 + jitter.
//   W[:, j]   = (K + s_j I)^-1 (y_j - mu)
//   lev[:, j] = 1 - sigma_j^2 diag((K + s_j I)^-1) = 1 - sigma_j^2 sum_k U_ik^2 / (lam_k + s_j)    :313-323,385-403
//   cr2       = (y - mu - K W)^2 / (1 - lev)^2                                                     :330-333
//   VW[:, j]  = (K + s_j I)^-1 cr2[:, j]                                                           :338-350
extern "C" int mln_full_conditional_noise(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n,
                                          int32_t d, const double* y, int64_t p, double mu, const double* sigma,
                                          double jitter, double* W, double* leverage, double* corrected_r2,
                                          double* variance_W) {
  if (!ctx || !x || !y || !sigma || !W) return MLN_ERR_ARG;
  if (n < 1 || n > 32768 || p < 1 || d < 1) { mln_set_error(ctx, "bad shape"); return MLN_ERR_SHAPE; }
  if ((corrected_r2 || variance_W) && !leverage) { mln_set_error(ctx, "corrected_r2 / variance_W need the leverage output"); return MLN_ERR_ARG; }
  if (variance_W && !corrected_r2) { mln_set_error(ctx, "variance_W needs the corrected_r2 output"); return MLN_ERR_ARG; }
  if (ctx->n_ranks > 1) { mln_set_error(ctx, "the full (non-sparse) GP cannot be cell-sharded"); return MLN_ERR_UNSUPPORTED; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevCov dc;
  MLN_TRY(mln_lower_cov(ctx, cov, d, &dc));
  DevIn dx, dy;
  MLN_TRY(dx.init(ctx, x, (size_t)n * d));
  MLN_TRY(dy.init(ctx, y, (size_t)n * p));
  DevOut oW, oH, oC, oV;
  MLN_TRY(oW.init(ctx, W, (size_t)n * p));
  if (leverage) MLN_TRY(oH.init(ctx, leverage, (size_t)n * p));
  if (corrected_r2) MLN_TRY(oC.init(ctx, corrected_r2, (size_t)n * p));
  if (variance_W) MLN_TRY(oV.init(ctx, variance_W, (size_t)n * p));
  const int64_t ld = pad16(n), np_ = n * p;
  double *K = nullptr, *V = nullptr, *D = nullptr, *T = nullptr, *d_lam = nullptr, *d_s = nullptr, *d_sig2 = nullptr;
  int rc = MLN_OK;
  auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == MLN_OK) rc = mln_hip_fail(ctx, e, "full_conditional_noise", __FILE__, __LINE__); };
  const size_t nn = sizeof(double) * (size_t)n * ld;
  chk(mln_dmalloc((void**)&K, nn));
  chk(mln_dmalloc((void**)&V, nn));
  chk(mln_dmalloc((void**)&D, sizeof(double) * (size_t)np_));
  chk(mln_dmalloc((void**)&T, sizeof(double) * (size_t)np_));
  chk(mln_dmalloc((void**)&d_lam, sizeof(double) * (size_t)n));
  chk(mln_dmalloc((void**)&d_s, sizeof(double) * (size_t)p));
  chk(mln_dmalloc((void**)&d_sig2, sizeof(double) * (size_t)p));
  if (rc == MLN_OK) chk(hipMemsetAsync(K, 0, nn, ctx->stream));
  if (rc == MLN_OK) rc = launch_kernel_matrix(ctx, dc, dx.dev, n, dx.dev, n, d, K, ld, 0.0);
  std::vector<double> lam((size_t)n), s((size_t)p), sig2((size_t)p);
  int sweeps = 0;
  if (rc == MLN_OK) rc = dev_eigh(ctx, K, n, ld, lam.data(), V, ld, &sweeps);        // row k of V = eigenvector k
  if (rc == MLN_OK) {
    double smin = 0.0;
    for (int64_t j = 0; j < p; ++j) {
      sig2[(size_t)j] = sigma[j] * sigma[j];
      s[(size_t)j] = sig2[(size_t)j] + jitter;
      if (j == 0 || s[(size_t)j] < smin) smin = s[(size_t)j];
    }
    if (!(lam[0] + smin > 0.0)) {
      mln_set_error(ctx, "Covariance not positively definite with the given sigma and jitter");
      rc = MLN_ERR_NOT_PD;
    }
  }
  auto resolvent = [&](const double* rhs, double* out) -> int {      // out = U diag(1/(lam + s_j)) U^T rhs, per column j
    GemmArgs g{};
    g.A = V; g.lda = ld; g.B = rhs; g.ldb = p; g.C = T; g.ldc = p;
    g.M = n; g.N = p; g.K = n; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 0; g.split_k = 1;
    int r = launch_dgemm(ctx, g);
    if (r != MLN_OK) return r;
    hipLaunchKernelGGL(k_mul_inplace, dim3(grid_1d(np_)), dim3(256), 0, ctx->stream, T, D, np_);
    g.B = T; g.C = out; g.ta = 1;
    return launch_dgemm(ctx, g);
  };
  if (rc == MLN_OK) {
    chk(hipMemcpyAsync(d_lam, lam.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    chk(hipMemcpyAsync(d_s, s.data(), sizeof(double) * (size_t)p, hipMemcpyHostToDevice, ctx->stream));
    chk(hipMemcpyAsync(d_sig2, sig2.data(), sizeof(double) * (size_t)p, hipMemcpyHostToDevice, ctx->stream));
    int64_t bx = (p + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_resolvent_table, dim3((unsigned)bx, (unsigned)n, 1u), dim3(256), 0, ctx->stream, D, p, n, p, d_lam, d_s);
    chk(hipGetLastError());
  }
  // weights: R = y - mu in oW, then the resolvent
  double* R = nullptr;
  chk(mln_dmalloc((void**)&R, sizeof(double) * (size_t)np_));
  if (rc == MLN_OK) hipLaunchKernelGGL(k_shift, dim3(grid_1d(np_)), dim3(256), 0, ctx->stream, R, dy.dev, mu, np_);
  if (rc == MLN_OK) rc = resolvent(R, oW.dev);
  if (rc == MLN_OK && leverage) {
    // Q = V o V (in place: V is not needed unsquared again until the variance solve, which re-reads K's eigenvectors
    // from a copy), H = Q^T D, then 1 - sigma^2 H
    double* Q = nullptr;
    chk(mln_dmalloc((void**)&Q, nn));
    if (rc == MLN_OK) chk(hipMemcpyAsync(Q, V, nn, hipMemcpyDeviceToDevice, ctx->stream));
    if (rc == MLN_OK) hipLaunchKernelGGL(k_square_inplace, dim3(grid_1d(n * ld)), dim3(256), 0, ctx->stream, Q, n * ld);
    GemmArgs g{};
    g.A = Q; g.lda = ld; g.B = D; g.ldb = p; g.C = oH.dev; g.ldc = p;
    g.M = n; g.N = p; g.K = n; g.alpha = 1.0; g.beta = 0.0; g.ta = 1; g.tb = 0; g.split_k = 1;
    if (rc == MLN_OK) rc = launch_dgemm(ctx, g);
    if (rc == MLN_OK) hipLaunchKernelGGL(k_leverage_finish, dim3(grid_1d(np_)), dim3(256), 0, ctx->stream, oH.dev, np_, p, d_sig2);
    (void)hipStreamSynchronize(ctx->stream);
    if (Q) (void)mln_dfree(Q);
  }
  if (rc == MLN_OK && corrected_r2) {
    GemmArgs g{};
    g.A = K; g.lda = ld; g.B = oW.dev; g.ldb = p; g.C = R; g.ldc = p;                 // K W (K is symmetric, kept by dev_eigh)
    g.M = n; g.N = p; g.K = n; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 0; g.split_k = 1;
    rc = launch_dgemm(ctx, g);
    if (rc == MLN_OK) hipLaunchKernelGGL(k_hc3, dim3(grid_1d(np_)), dim3(256), 0, ctx->stream, oC.dev, dy.dev, R, mu, oH.dev, np_);
    if (rc == MLN_OK && variance_W) rc = resolvent(oC.dev, oV.dev);                    // variance_mu = 0
  }
  chk(hipGetLastError());
  if (rc == MLN_OK) rc = oW.commit();
  if (rc == MLN_OK && leverage) rc = oH.commit();
  if (rc == MLN_OK && corrected_r2) rc = oC.commit();
  if (rc == MLN_OK && variance_W) rc = oV.commit();
  (void)hipStreamSynchronize(ctx->stream);
  void* ptrs[] = {K, V, D, T, R, d_lam, d_s, d_sig2};
  for (void* q : ptrs) if (q) (void)mln_dfree(q);
  return rc;
}

extern "C" int mln_sparse_solve_noise(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local,
                                      int32_t d, const double* xu, int64_t m, const double* y, int64_t p, double mu,
                                      const double* sigma, int32_t sigma_kind, double jitter, double* W) {
  return sparse_solve_impl(ctx, cov, x, n_local, d, xu, m, y, p, mu, sigma, sigma_kind, jitter, W, nullptr, nullptr);
}

