// Fused predictive mean, persistent-row form (reference: conditional.py:899-906 `_mean`, one output column): the
// dispatcher over the per-kind translation units predict_rows_*.hip (kernel: predict_rows_impl.h).
#include "cov_rows.h"

#define MLN_PREDICT_ROWS_DECL(NAME)                                                                                    \
  int NAME(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m, int d,             \
           const double* xx, const double* yy, const double* w, double mu, double* out);
MLN_PREDICT_ROWS_DECL(launch_predict_mean_rows_matern32)
MLN_PREDICT_ROWS_DECL(launch_predict_mean_rows_matern52)
MLN_PREDICT_ROWS_DECL(launch_predict_mean_rows_expquad)
MLN_PREDICT_ROWS_DECL(launch_predict_mean_rows_exponential)
MLN_PREDICT_ROWS_DECL(launch_predict_mean_rows_ratquad)
#undef MLN_PREDICT_ROWS_DECL

int launch_predict_mean_rows(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m,
                             int d, const double* xx, const double* yy, const double* w, double mu, double* out) {
  switch (cov.leaves[0].kind) {
    case MLN_K_MATERN32: return launch_predict_mean_rows_matern32(ctx, cov, x, n, y, m, d, xx, yy, w, mu, out);
    case MLN_K_MATERN52: return launch_predict_mean_rows_matern52(ctx, cov, x, n, y, m, d, xx, yy, w, mu, out);
    case MLN_K_EXPQUAD: return launch_predict_mean_rows_expquad(ctx, cov, x, n, y, m, d, xx, yy, w, mu, out);
    case MLN_K_EXPONENTIAL: return launch_predict_mean_rows_exponential(ctx, cov, x, n, y, m, d, xx, yy, w, mu, out);
    default: return launch_predict_mean_rows_ratquad(ctx, cov, x, n, y, m, d, xx, yy, w, mu, out);
  }
}
