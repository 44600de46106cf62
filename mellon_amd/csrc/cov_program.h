// Forward-mode differentiation of the postfix covariance program, shared by cov_grad.hip and cov_kernels.hip.
#pragma once
#include "mln_internal.h"

static __device__ __forceinline__ double pick4(const double v[MLN_MAX_LEAVES], int id) {
  double r = v[0];
#pragma unroll
  for (int l = 1; l < MLN_MAX_LEAVES; ++l) r = (id == l) ? v[l] : r;
  return r;
}

// a[l] = dP/dk_l by forward-mode evaluation of the postfix program (base_cov.py:341-364 Add,
// :407-438 Mul, :481-497 Pow with a scalar exponent).
static __device__ __forceinline__ void program_adjoints(const DevCov& cov, const double kv[MLN_MAX_LEAVES],
                                                 double a[MLN_MAX_LEAVES]) {
#pragma unroll
  for (int l = 0; l < MLN_MAX_LEAVES; ++l) {
    a[l] = 0.0;
    if (l >= cov.n_leaves) continue;
    double v0 = 0.0, t0 = 0.0, v1 = 0.0, t1 = 0.0, v2 = 0.0, t2 = 0.0;   // top, below, below
    for (int t = 0; t < cov.n_toks; ++t) {
      const int op = cov.tok_op[t];
      if (op == MLN_OP_LEAF || op == MLN_OP_CONST) {
        v2 = v1; t2 = t1; v1 = v0; t1 = t0;
        if (op == MLN_OP_CONST) { v0 = cov.tok_val[t]; t0 = 0.0; }
        else { const int id = cov.tok_leaf[t]; v0 = pick4(kv, id); t0 = (id == l) ? 1.0 : 0.0; }
      } else {
        const double lv = v1, lt = t1, rv = v0, rt = t0;
        if (op == MLN_OP_ADD) { v0 = lv + rv; t0 = lt + rt; }
        else if (op == MLN_OP_MUL) { v0 = lv * rv; t0 = lt * rv + lv * rt; }
        else { v0 = pow(lv, rv); t0 = rv * pow(lv, rv - 1.0) * lt; }
        v1 = v2; t1 = t2;
      }
    }
    a[l] = t0;
  }
}

