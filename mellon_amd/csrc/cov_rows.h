// Shared pieces of the persistent-row covariance kernels (cov_rows.hip, predict_rows.hip, the 1-NN kernel).
#pragma once
#include "mln_core.h"

typedef double v4d_t __attribute__((ext_vector_type(4)));

namespace covrows {
constexpr int TN = 64;    // centres per tile
constexpr int NNS = 65;   // LDS row stride (doubles), odd: the 16 lanes of one ds_read2_b64 group (li = 0..15, one k) land on 16 distinct bank pairs
}  // namespace covrows

// persistent-row launchers: one stationary leaf over all d <= 64 contiguous columns
int launch_kernel_matrix_rows(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m,
                              int d, const double* xx, const double* yy, double* out, int64_t ldo, double add_diag,
                              float* out32);
int launch_predict_mean_rows(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m,
                             int d, const double* xx, const double* yy, const double* w, double mu, double* out);
