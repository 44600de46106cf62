// Dispatcher of the persistent-row kernel-matrix pass (kernels: cov_rows_impl.h, one translation unit per kind).
#include "cov_rows.h"
#include "cov_rows_q.h"

#define MLN_ROWS_DECL(NAME)                                                                                     \
  int NAME(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m, int d,      \
           const double* xx, const double* yy, double* out, int64_t ldo, double add_diag, float* out32, int q32);
MLN_ROWS_DECL(launch_kernel_matrix_rows_matern32)
MLN_ROWS_DECL(launch_kernel_matrix_rows_matern52)
MLN_ROWS_DECL(launch_kernel_matrix_rows_expquad)
MLN_ROWS_DECL(launch_kernel_matrix_rows_exponential)
#undef MLN_ROWS_DECL

int launch_kernel_matrix_rows(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m,
                              int d, const double* xx, const double* yy, double* out, int64_t ldo, double add_diag,
                              float* out32) {
  return launch_kernel_matrix_rows_q(ctx, cov, x, n, y, m, d, xx, yy, out, ldo, add_diag, out32, 0);
}

int launch_kernel_matrix_rows_q(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m,
                                int d, const double* xx, const double* yy, double* out, int64_t ldo, double add_diag,
                                float* out32, int q32) {
  switch (cov.leaves[0].kind) {
    case MLN_K_MATERN32: return launch_kernel_matrix_rows_matern32(ctx, cov, x, n, y, m, d, xx, yy, out, ldo, add_diag, out32, q32);
    case MLN_K_MATERN52: return launch_kernel_matrix_rows_matern52(ctx, cov, x, n, y, m, d, xx, yy, out, ldo, add_diag, out32, q32);
    case MLN_K_EXPQUAD: return launch_kernel_matrix_rows_expquad(ctx, cov, x, n, y, m, d, xx, yy, out, ldo, add_diag, out32, q32);
    case MLN_K_EXPONENTIAL: return launch_kernel_matrix_rows_exponential(ctx, cov, x, n, y, m, d, xx, yy, out, ldo, add_diag, out32, q32);
    default:
      mln_set_error(ctx, "persistent-row kernel matrix: unsupported leaf kind");   // RatQuad (pow) takes the tiled kernel
      return MLN_ERR_UNSUPPORTED;
  }
}
