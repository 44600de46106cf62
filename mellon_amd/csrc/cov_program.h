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


// a1[l] = dP/dk_l and a2[l][l'] = d2P/dk_l dk_l' (l <= l') by hyper-dual evaluation of the postfix program
// (value, d/dk_l, d/dk_l', d2/dk_l dk_l'); Pow takes a constant exponent as in program_adjoints.
static __device__ __forceinline__ void program_second(const DevCov& cov, const double kv[MLN_MAX_LEAVES],
                                                      double a1[MLN_MAX_LEAVES],
                                                      double a2[MLN_MAX_LEAVES][MLN_MAX_LEAVES]) {
#pragma unroll
  for (int l = 0; l < MLN_MAX_LEAVES; ++l) {
    a1[l] = 0.0;
#pragma unroll
    for (int lp = 0; lp < MLN_MAX_LEAVES; ++lp) a2[l][lp] = 0.0;
  }
#pragma unroll
  for (int l = 0; l < MLN_MAX_LEAVES; ++l) {
#pragma unroll
    for (int lp = l; lp < MLN_MAX_LEAVES; ++lp) {
      if (lp >= cov.n_leaves) continue;
      double v[3] = {0.0, 0.0, 0.0}, p[3] = {0.0, 0.0, 0.0}, q[3] = {0.0, 0.0, 0.0}, pq[3] = {0.0, 0.0, 0.0};   // [0] = top
      for (int t = 0; t < cov.n_toks; ++t) {
        const int op = cov.tok_op[t];
        if (op == MLN_OP_LEAF || op == MLN_OP_CONST) {
          v[2] = v[1]; p[2] = p[1]; q[2] = q[1]; pq[2] = pq[1];
          v[1] = v[0]; p[1] = p[0]; q[1] = q[0]; pq[1] = pq[0];
          if (op == MLN_OP_CONST) { v[0] = cov.tok_val[t]; p[0] = 0.0; q[0] = 0.0; }
          else {
            const int id = cov.tok_leaf[t];
            v[0] = pick4(kv, id); p[0] = (id == l) ? 1.0 : 0.0; q[0] = (id == lp) ? 1.0 : 0.0;
          }
          pq[0] = 0.0;
        } else {
          const double av = v[1], ap = p[1], aq = q[1], apq = pq[1], bv = v[0], bp = p[0], bq = q[0], bpq = pq[0];
          if (op == MLN_OP_ADD) { v[0] = av + bv; p[0] = ap + bp; q[0] = aq + bq; pq[0] = apq + bpq; }
          else if (op == MLN_OP_MUL) {
            v[0] = av * bv; p[0] = ap * bv + av * bp; q[0] = aq * bv + av * bq;
            pq[0] = apq * bv + ap * bq + aq * bp + av * bpq;
          } else {
            const double f1 = bv * pow(av, bv - 1.0), f2 = bv * (bv - 1.0) * pow(av, bv - 2.0);
            v[0] = pow(av, bv); p[0] = f1 * ap; q[0] = f1 * aq; pq[0] = f2 * ap * aq + f1 * apq;
          }
          v[1] = v[2]; p[1] = p[2]; q[1] = q[2]; pq[1] = pq[2];
        }
      }
      a2[l][lp] = pq[0];
      if (lp == l) a1[l] = p[0];
    }
  }
}
