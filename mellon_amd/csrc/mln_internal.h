// Internal declarations shared by the translation units of libmellon_hip.so (gfx950 only).
#pragma once
#include "mln_core.h"

// comm.hip: the collectives of the sharded path, over RCCL or the in-process loopback group
int comm_allreduce(mln_ctx* ctx, double* dev, int64_t count);                       // sum, in place, same bits on every rank
int comm_bcast0(mln_ctx* ctx, double* dev, int64_t count);                          // rank 0 -> all
int comm_allgather(mln_ctx* ctx, const double* send, double* recv, int64_t count);  // count per rank, rank order
void comm_release(mln_ctx* ctx);

// ---- kernel launchers (all asynchronous on ctx->stream; device pointers only) ---------------
// cov_kernels.hip
// out32 (optional): a 32-bit copy of the result for the warm-up passes of the MAP solve -- fp32 values, or with
// q32 != 0 the 32-bit fixed-point number round(v 2^32) (covariance values in [0, 1]: absolute error 1.2e-10
// everywhere, where fp32 has 3e-8 near 1)
int launch_kernel_matrix(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y,
                         int64_t m, int d, double* out, int64_t ldo, double add_diag, float* out32 = nullptr,
                         int q32 = 0);
__device__ __forceinline__ float mln_surrogate_bits(double v, int q32) {
  const unsigned fx = (unsigned)fmin(fma(v, 4294967296.0, 0.5), 4294967295.0);
  return q32 ? __uint_as_float(fx) : (float)v;
}
int launch_predict_mean1(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y,
                         int64_t m, int d, const double* w, double mu, double* out);

int launch_cov_diag(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, int d, double* out);
int launch_row_sumsq(mln_ctx* ctx, const double* T, int64_t ld, int64_t rows, int64_t cols, const double* base,
                     double sign, double* out);
int launch_nn_distances(mln_ctx* ctx, const double* x, int64_t n, const double* y, int64_t m, int d,
                        int64_t self_offset, double* out);

// dgemm.hip : C = alpha * op(A) op(B) + beta * C   (row-major, fp64 MFMA 16x16x4)
//   ta = 0: A is M x K (lda >= K);  ta = 1: A is stored K x M (lda >= M)
//   tb = 0: B is K x N (ldb >= N);  tb = 1: B is stored N x K (ldb >= K)
//   lower_only: 1 = skip tiles strictly above the diagonal (square C); 2 = also skip the diagonal tiles;
//               3 = skip the tiles strictly BELOW the diagonal
//   kmode:      1 = the K range of a tile is its own ROW block [m0, m0 + 128) -- op(A) block-diagonal;
//               2 = its own COLUMN block [n0, n0 + 128) -- op(B) block-diagonal   (128 x 128 tiles)
//               3 = [0, m0 + tile): op(A) lower triangular;  4 = [0, n0 + tile): op(B) upper triangular;
//               7 = [n0, m0 + tile): the product of two lower triangular matrices (its lower tiles)
//   split_k > 1: writes split_k partial C's at C + s * c_split_stride (beta ignored, alpha applied)
struct GemmArgs {
  const double* A; int64_t lda;
  const double* B; int64_t ldb;
  double* C; int64_t ldc;
  int64_t M, N, K;
  double alpha, beta;
  int ta, tb;
  int lower_only;
  int split_k; int64_t c_split_stride;
  int kmode;
  // batch > 1: that many independent products in ONE launch (grid z); operand b sits bsa / bsb / bsc doubles after operand 0
  // (any sign: two separately allocated matrices are a batch of two).  Same shapes, same leading dimensions.
  int batch; int64_t bsa, bsb, bsc;
};
int launch_dgemm(mln_ctx* ctx, const GemmArgs& g);
int launch_sum_partials(mln_ctx* ctx, const double* parts, int n_parts, int64_t stride, double* out,
                        int64_t count, double beta);

// linalg.hip (block-solve helpers are declared in linalg.h)
int dev_cholesky_lower(mln_ctx* ctx, double* A, int64_t m, int64_t lda);       // in place; zeroes upper
// the same for TWO matrices of one shape in one chain of launches (A2 may be anywhere: the pointer difference is the batch
// stride); *bad (may be NULL): bit 0 / bit 1 set when the first / second matrix hit a non-positive pivot (MLN_ERR_NOT_PD then)
int dev_cholesky_lower2(mln_ctx* ctx, double* A, double* A2, int64_t m, int64_t lda, int* bad);
int launch_add_diag(mln_ctx* ctx, double* A, int64_t m, int64_t lda, double v);
int launch_symmetrize_from_lower(mln_ctx* ctx, double* A, int64_t m, int64_t lda);
// gram_i8.hip: lower 128-tiles of alpha * Q^T Q, Q = round(K 8355711) in three int8 digit planes, per k-chunk (`n_splits`
// partial results `part_stride` doubles apart, to be summed by the caller); K holds values in [0, 1], row pitch ldk
// tridiag.hip: A (m x m symmetric, full storage, destroyed) -> number of eigenvalues above tol2 * lambda_max
int dev_sym_rank_above(mln_ctx* ctx, double* A, int64_t m, int64_t ld, double tol2, int64_t* rank, double* lambda_max);
// ldl_inertia.hip: the same count from the inertia of A - x I (A destroyed); *ok = false: not certified, use the above
int dev_sym_rank_above_ldl(mln_ctx* ctx, double* A, int64_t m, int64_t ld, double tol2, int64_t* rank, double* lambda_max, bool* ok);
int gram_i8_splits(int64_t rows, int64_t m);
int launch_gram_i8(mln_ctx* ctx, const double* K, int64_t ldk, int64_t rows, int64_t m, double alpha, double* parts,
                   int64_t ldg, int64_t part_stride, int n_splits);
int launch_axpby(mln_ctx* ctx, int64_t n, double a, const double* x, double b, double* y);

// cov_grad.hip
int launch_kernel_grad(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y, int64_t m,
                       int d, int exact_denominator, double* out);
int launch_predict_gradient_gemm(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* c,
                                 int64_t m, int d, const double* w, double* out);   // single stationary leaf
int launch_predict_gradient_gemm_multi(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* c,
                                       int64_t m, int d, const double* w, double* out);   // stationary leaves, any program
int launch_predict_hessian(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* c, int64_t m,
                           int d, const double* w, double* out);              // out: n x d x d
int launch_predict_gradient(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* c, int64_t m,
                            int d, const double* w, double* out);

// eigh.hip: symmetric eigensolver (block one-sided Jacobi); w_host ascending, Vrows row j = eigenvector j
int dev_eigh(mln_ctx* ctx, const double* A, int64_t m, int64_t lda, double* w_host, double* Vrows, int64_t ldv,
             int* n_sweeps_out);

