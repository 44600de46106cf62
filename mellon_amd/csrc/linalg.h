// Block-scaled triangular factors for GEMM-based triangular solves (see linalg.hip).
#pragma once
#include "mln_internal.h"

struct TriInv {
  double* W = nullptr;   // row-scaled    (forward solves, X Lf^-T)
  double* W2 = nullptr;  // column-scaled (transposed / backward solves)
  int64_t m = 0, ld = 0;
  const double* Lf = nullptr;  // the factor itself (not owned; must outlive this object)
  int64_t ldf = 0;
};

int triinv_build(mln_ctx* ctx, const double* Lf, int64_t m, int64_t ld, bool need_w, bool need_w2, TriInv* out);
void triinv_free(TriInv* t);
int triinv_solve_right_T(mln_ctx* ctx, const TriInv& t, double* X, int64_t n, int64_t ldx);  // X <- X Lf^-T
// `tri_b`: B is itself lower triangular (solve_left) / upper triangular (solve_left_T) with m columns: the result has
// the same shape and only the columns that can be non-zero in each row block are computed (half the flops).
int triinv_solve_left(mln_ctx* ctx, const TriInv& t, double* B, int64_t p, int64_t ldb, bool tri_b = false);     // B <- Lf^-1 B
int triinv_solve_left_T(mln_ctx* ctx, const TriInv& t, double* B, int64_t p, int64_t ldb, bool tri_b = false);   // B <- Lf^-T B
int launch_copy_block(mln_ctx* ctx, const double* src, int64_t lds, double* dst, int64_t ldd, int64_t rows,
                      int64_t cols);
int launch_transpose(mln_ctx* ctx, const double* src, int64_t lds, double* dst, int64_t ldd, int64_t m);  // dst = src^T (m x m)
// potrf.hip: Cholesky of one nb x nb (nb <= 128) diagonal block in place + the inverse of its factor (Dinv: 128 x 128,
// leading dimension 128, strictly-upper blocks untouched); *info = global pivot index + 1 on a bad pivot
// nbatch > 1: the same block of nbatch matrices a_bs doubles apart (Dinv: nbatch slabs of 128 x 128, info: nbatch words)
int launch_potrf128(mln_ctx* ctx, double* A, int64_t lda, int nb, double* Dinv, int* info, int64_t j0, int nbatch = 1, int64_t a_bs = 0);

