// predict_rows_prod.hip: the persistent-row fused predictive mean for the time-sensitive product kernel (state leaf x time
// leaf).  (Its own header: cov_rows.h is included by the slow-to-compile persistent-row translation units.)
#pragma once
#include "mln_core.h"
bool predict_rows_prod_eligible(const DevCov& cov, int d);
int launch_predict_mean_rows_prod(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m,
                                  int d, const double* xx0, const double* yy0, const double* w, double mu, double* out);
// kernel_rows_prod_*.hip: the kernel-matrix pass of the same product (one translation unit per kind), and its dispatcher
#define MLN_DECLARE_ROWS_PROD(NAME)                                                                                     \
  int NAME(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m, int d,              \
           const double* xx0, const double* yy0, double* out, int64_t ldo, double add_diag, float* out32, int q32);
MLN_DECLARE_ROWS_PROD(launch_kernel_matrix_rows_prod_matern32)
MLN_DECLARE_ROWS_PROD(launch_kernel_matrix_rows_prod_matern52)
MLN_DECLARE_ROWS_PROD(launch_kernel_matrix_rows_prod_expquad)
MLN_DECLARE_ROWS_PROD(launch_kernel_matrix_rows_prod_exponential)
MLN_DECLARE_ROWS_PROD(launch_kernel_matrix_rows_prod)
#undef MLN_DECLARE_ROWS_PROD
