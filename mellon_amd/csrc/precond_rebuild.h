// Importance-sampled rows for the solver's second preconditioner (precond_rebuild.hip).
#pragma once
#include "mln_core.h"

struct RebuildSelection {
  int64_t rows;      // selected rows of THIS rank
  int64_t* idx;      // their local indices, ascending (device)
  double* scale;     // sqrt(w_i / w_max) per selected row (device)
  double w_max;      // global; the Gram of the scaled rows times w_max estimates sum_i a_i L_i L_i^T
  double c;          // p_i = min(1, c a_i)
  double sum_a;      // global sum of the weights
};

// f_dev, V_dev: n rows of this rank (f = L z + mu at the solver's accepted point); row0: global index of its first cell.
// Collective: every rank calls it (all-reduces of the weight sums).
int rebuild_select_rows(mln_ctx* ctx, const double* f_dev, const double* V_dev, int64_t n, int64_t row0,
                        double target_rows_global, uint64_t seed, RebuildSelection* out, double cap = 1e300);
// (cap: weights are e^{min(f + V, cap)} -- the second derivative of the solver's capped likelihood term)
void rebuild_selection_free(mln_ctx* ctx, RebuildSelection* s);
int launch_gather_scale_rows(mln_ctx* ctx, const double* A, int64_t ld, const int64_t* idx, const double* scale,
                             int64_t rows, double* R);
