// The time-sensitive product kernel  k(ls, active_dims=:-1) * k(ls_time, active_dims=-1)  in the persistent-row kernels
// (reference: parameters.py:641-644): eligibility test and the dispatchers over the per-kind translation units
// predict_rows_prod_*.hip (fused predictive mean, predict_rows_prod_impl.h) and kernel_rows_prod_*.hip (kernel matrix).
#include "cov_rows.h"
#include "predict_rows_prod.h"

// The program  leaf0 leaf1 MUL  with leaf0 over columns 0 .. d-2, leaf1 over column d-1, both of the same stationary kind
bool predict_rows_prod_eligible(const DevCov& cov, int d) {
  if (cov.n_leaves != 2 || cov.n_toks != 3 || d < 2 || d > 65) return false;
  if (cov.tok_op[0] != MLN_OP_LEAF || cov.tok_op[1] != MLN_OP_LEAF || cov.tok_op[2] != MLN_OP_MUL) return false;
  if (cov.tok_leaf[0] != 0 || cov.tok_leaf[1] != 1) return false;
  const DevLeaf &l0 = cov.leaves[0], &l1 = cov.leaves[1];
  if (l0.kind != l1.kind || l0.kind < MLN_K_MATERN32 || l0.kind > MLN_K_EXPONENTIAL) return false;
  if (l0.ndims != d - 1 || l1.ndims != 1 || cov.dims[l1.dims_off] != d - 1) return false;
  for (int k = 0; k < d - 1; ++k) if (cov.dims[l0.dims_off + k] != k) return false;
  return true;
}

#define MLN_PP_DECL(NAME)                                                                                              \
  int NAME(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m, int d,             \
           const double* xx0, const double* yy0, const double* w, double mu, double* out);
MLN_PP_DECL(launch_predict_mean_rows_prod_matern32)
MLN_PP_DECL(launch_predict_mean_rows_prod_matern52)
MLN_PP_DECL(launch_predict_mean_rows_prod_expquad)
MLN_PP_DECL(launch_predict_mean_rows_prod_exponential)
#undef MLN_PP_DECL

int launch_predict_mean_rows_prod(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m,
                                  int d, const double* xx0, const double* yy0, const double* w, double mu, double* out) {
  switch (cov.leaves[0].kind) {
    case MLN_K_MATERN32: return launch_predict_mean_rows_prod_matern32(ctx, cov, x, n, y, m, d, xx0, yy0, w, mu, out);
    case MLN_K_MATERN52: return launch_predict_mean_rows_prod_matern52(ctx, cov, x, n, y, m, d, xx0, yy0, w, mu, out);
    case MLN_K_EXPQUAD: return launch_predict_mean_rows_prod_expquad(ctx, cov, x, n, y, m, d, xx0, yy0, w, mu, out);
    default: return launch_predict_mean_rows_prod_exponential(ctx, cov, x, n, y, m, d, xx0, yy0, w, mu, out);
  }
}

int launch_kernel_matrix_rows_prod(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m,
                                   int d, const double* xx0, const double* yy0, double* out, int64_t ldo, double add_diag,
                                   float* out32, int q32) {
  switch (cov.leaves[0].kind) {
    case MLN_K_MATERN32: return launch_kernel_matrix_rows_prod_matern32(ctx, cov, x, n, y, m, d, xx0, yy0, out, ldo, add_diag, out32, q32);
    case MLN_K_MATERN52: return launch_kernel_matrix_rows_prod_matern52(ctx, cov, x, n, y, m, d, xx0, yy0, out, ldo, add_diag, out32, q32);
    case MLN_K_EXPQUAD: return launch_kernel_matrix_rows_prod_expquad(ctx, cov, x, n, y, m, d, xx0, yy0, out, ldo, add_diag, out32, q32);
    case MLN_K_EXPONENTIAL: return launch_kernel_matrix_rows_prod_exponential(ctx, cov, x, n, y, m, d, xx0, yy0, out, ldo, add_diag, out32, q32);
    default: break;
  }
  mln_set_error(ctx, "kernel_matrix_rows_prod: unsupported kind");
  return MLN_ERR_UNSUPPORTED;
}
