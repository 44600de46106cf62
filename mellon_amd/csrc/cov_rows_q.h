// The persistent-row kernel-matrix launcher with the format of the 32-bit copy (kept out of cov_rows.h, which the
// slow-to-compile predict_rows.hip includes).
#pragma once
#include "mln_core.h"
int launch_kernel_matrix_rows_q(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m,
                                int d, const double* xx, const double* yy, double* out, int64_t ldo, double add_diag,
                                float* out32, int q32);
