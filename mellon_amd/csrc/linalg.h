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
int launch_potrf128(mln_ctx* ctx, double* A, int64_t lda, int nb, double* Dinv, int* info, int64_t j0);

// CU-masked side streams (linalg.hip).  Kernels of two streams only run side by side on this GPU when neither fills the
// dispatcher (round 1: a second stream next to the 7813-workgroup kernel-matrix pass gained nothing).  A stream whose CU
// mask leaves `free_cus` compute units out guarantees room for whatever the context's own (unmasked) stream launches
// meanwhile.  Streams are created once per (context, free_cus) and live as long as the context; nullptr when the runtime
// refuses (callers then keep everything on ctx->stream).  lookahead_disabled(): thread-local switch for a caller that
// already shares the GPU with a long kernel (fit_prepare's landmark chain).
hipStream_t masked_stream(mln_ctx* ctx, int free_cus);
hipEvent_t masked_stream_event(mln_ctx* ctx, int which);     // four reusable events per context (timing disabled)
void masked_streams_release(mln_ctx* ctx);
void set_lookahead_disabled(bool off);
