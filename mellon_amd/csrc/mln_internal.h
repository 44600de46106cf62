// Internal declarations shared by the translation units of libmellon_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/mellon_hip.h"

struct mln_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int n_cu = 0;
  void* comm = nullptr;  // ncclComm_t when multi-GPU
  int n_ranks = 1;
  int rank = 0;
  std::string err;
  // grow-only device scratch
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  int* d_info = nullptr;  // device int[4] for factorisation status
};

void mln_set_error(mln_ctx* ctx, const std::string& msg);
int mln_hip_fail(mln_ctx* ctx, hipError_t e, const char* what, const char* file, int line);

#define MLN_HIP(ctx, call)                                                   \
  do {                                                                       \
    hipError_t e__ = (call);                                                 \
    if (e__ != hipSuccess) return mln_hip_fail((ctx), e__, #call, __FILE__, __LINE__); \
  } while (0)

#define MLN_TRY(call)            \
  do {                           \
    int s__ = (call);            \
    if (s__ != MLN_OK) return s__; \
  } while (0)

// alloc.hip: caching device allocator (every internal device buffer goes through it)
hipError_t mln_dmalloc(void** out, size_t bytes);
hipError_t mln_dfree(void* p);
void mln_dcache_flush();

// ---- device-side covariance program (by-value kernel argument) ------------------------------
struct DevLeaf {
  int kind;
  int ndims;
  int dims_off;  // offset into DevCov::dims
  int pad;
  double ls;
  double alpha;
  double alpha_inv_ls[2];  // [1] = 1 / ls
};
struct DevCov {
  int n_leaves;
  int n_toks;
  DevLeaf leaves[MLN_MAX_LEAVES];
  int tok_op[MLN_MAX_TOKS];
  int tok_leaf[MLN_MAX_TOKS];
  double tok_val[MLN_MAX_TOKS];
  short dims[MLN_MAX_DIMS];
};
int mln_lower_cov(mln_ctx* ctx, const mln_kernel_desc* cov, int d, DevCov* out);

// ---- kernel launchers (all asynchronous on ctx->stream; device pointers only) ---------------
// cov_kernels.hip
int launch_kernel_matrix(mln_ctx* ctx, const DevCov& cov, const double* x, int64_t n, const double* y,
                         int64_t m, int d, double* out, int64_t ldo, double add_diag, float* out32 = nullptr);
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
//   lower_only: skip 128x128 tiles strictly above the diagonal (square C)
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
};
int launch_dgemm(mln_ctx* ctx, const GemmArgs& g);
int launch_sum_partials(mln_ctx* ctx, const double* parts, int n_parts, int64_t stride, double* out,
                        int64_t count, double beta);

// linalg.hip (block-solve helpers are declared in linalg.h)
int dev_cholesky_lower(mln_ctx* ctx, double* A, int64_t m, int64_t lda);       // in place; zeroes upper
int launch_add_diag(mln_ctx* ctx, double* A, int64_t m, int64_t lda, double v);
int launch_symmetrize_from_lower(mln_ctx* ctx, double* A, int64_t m, int64_t lda);
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
  const float* L32;       // if non-null: stream this fp32 copy of L instead (same shape / leading dimension)
};
int objective_max_m();
int launch_to_f32(mln_ctx* ctx, const double* src, float* dst, int64_t count);
int launch_objective(mln_ctx* ctx, const ObjArgs& a);
int launch_reduce_obj(mln_ctx* ctx, const ObjArgs& a, double* out_loss_grad /* 1 + m [+ m] */);
int launch_gemv_rows(mln_ctx* ctx, const double* M, int64_t ld, int64_t rows, int64_t cols, const double* x,
                     double* y);   // y = M x, one wave per row
int launch_gemv_rows_tri(mln_ctx* ctx, const double* M, int64_t ld, int64_t rows, const double* x, double* y,
                         int upper, int64_t blk, int64_t ncol, int64_t seg);   // triangular blocks: non-zero part only

// helpers (api.hip)
int mln_scratch(mln_ctx* ctx, size_t bytes, void** out);
