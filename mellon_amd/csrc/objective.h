// The streaming pass over the n x m factor and the small m x m products around it (objective.hip).
#pragma once
#include "mln_core.h"

// objective.hip
struct ObjArgs {
  const double* L; int64_t ldl; int64_t n; int64_t m;
  const double* z; const double* V; const double* Vdr; double mu;
  double* part_grad;   // n_wg x m_pad
  double* part_hess;   // n_wg x m_pad or null
  double* part_loss;   // n_wg
  const double* weights;  // if non-null: "gemv-T" mode, grad_j = sum_i weights_i L_ij (V, Vdr, z unused)
  double* f_out;          // if non-null: store f_i = L_i . z + mu
  int n_wg; int64_t m_pad;
  const float* L32;       // if non-null: stream this 32-bit copy of L instead (same shape / leading dimension)
  int l32_fixed;          //   its format: 0 = fp32 values, 1 = 32-bit fixed point (value = bits / 2^32)
  double* f_keep[2];      // if f_slot is non-null: f_i = L_i . z + mu of every row is also stored to f_keep[1 - *f_slot]
  const int* f_slot;      //   (the solver's "trial" buffer; accepting a point flips the slot -- the log-density at the
                          //    optimum then needs no pass of its own)
  const double* cap;      // if non-null (device): e^t is continued linearly beyond t = *cap (solver.hip "capped start")
  int* over_flag;         // if non-null (device): += the number of rows of this launch above the cap (> 0: the capped and the true
                          //   objective differ at this point); read and cleared by the reduction (launch_reduce_obj2 -> out_loss[1])
  const int* gate;        // if non-null: the launch is a no-op unless *gate == gate_want (device-resident solver:
  int gate_want;          //   MLN_GATE_F64 / MLN_GATE_F32 select the streamed copy, MLN_GATE_DONE stops everything)
  int64_t row_stride;     // > 1: the pass covers the rows row_first + i * row_stride, i < n (n = their number), of the
  int64_t row_first;      //   buffer and of V / Vdr / weights -- the subsample objective of the solver's first phase
  double out_scale;       // != 0: factor applied to this launch's loss / gradient partials (row_stride for that objective)
  const int* gate2;       // second condition of a gated launch: no-op unless *gate2 == gate2_want (the solver's subsample
  int gate2_want;         //   LEVEL: one strided launch per level is enqueued, the state picks which one works)
  int64_t seg_cols;       // > 0: L points at a segment of seg_cols columns of a wider matrix (row pitch ldl): m > 8192
  int64_t seg_left;       //      ... and this many (padded) columns remain in the row from that pointer
  int f_accum;            // f_out mode: add this segment's dot products to what f_out already holds
};
// (the objective kernels compare gate & 3 with gate_want: MLN_GATE_F32C streams the same copy as MLN_GATE_F32;
//  MLN_GATE_SUB selects the launch over the row subsample; MLN_GATE_PAUSE, like DONE, stops every launch of the chain
//  until the host has rebuilt the preconditioner and resumed the solver)
enum { MLN_GATE_F64 = 0, MLN_GATE_F32 = 1, MLN_GATE_DONE = 2, MLN_GATE_SUB = 3, MLN_GATE_F32C = 5, MLN_GATE_PAUSE = 6 };
int objective_max_m();
int objective_max_m_one_pass();   // beyond it the pass is segmented (launch_objective_wide): no row map, no device-resident solver
bool objective_can_keep_f(int64_t n, int n_wg);
int launch_to_f32(mln_ctx* ctx, const double* src, float* dst, int64_t count);
int launch_objective(mln_ctx* ctx, const ObjArgs& a);
int launch_reduce_obj(mln_ctx* ctx, const ObjArgs& a, double* out_loss_grad /* 1 + m [+ m] */);
// the same reduction with the loss and the gradient going to separate places (no-op when *a.gate == MLN_GATE_DONE)
int launch_reduce_obj2(mln_ctx* ctx, const ObjArgs& a, double* out_loss, double* out_grad);
int launch_gemv_rows(mln_ctx* ctx, const double* M, int64_t ld, int64_t rows, int64_t cols, const double* x,
                     double* y);   // y = M x, one wave per row
int launch_gemv_rows_tri(mln_ctx* ctx, const double* M, int64_t ld, int64_t rows, const double* x, double* y,
                         int upper, int64_t blk, int64_t ncol, int64_t seg);   // triangular blocks: non-zero part only
// General form: rows >= blk write to y2[r - blk] (when y2 != null); the second column segment starts at column
// `mseg` of M and at element `xseg` of x; no-op when *gate == MLN_GATE_DONE.
struct GemvTri {
  const double* M; int64_t ld; int64_t rows;
  const double* x; double* y; double* y2;
  int upper;                 // 0: lower triangular blocks, 1: upper triangular blocks, 2: full rows (columns [0, ncol))
  int64_t blk, ncol, mseg, xseg;
  const int* gate;
  const double* xadd;        // if given: the product is taken with x + xadd (first segment)
};
int launch_gemv_tri(mln_ctx* ctx, const GemvTri& g);

