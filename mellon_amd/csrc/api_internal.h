// Shared by the translation units behind the C ABI (api*.hip): the fit handle, the host/device pointer wrappers and the
// helpers that cross the file boundaries.  (Round 4: csrc/api.hip -- 3000 lines -- split by concern:
//   api.hip          context, memory, stand-alone operators (kernel matrix, gradients, GEMM, Cholesky, eigh), prediction
//   api_fit.hip      the fit handle: Lp, the n x m kernel matrix, values-in route, Gram spectrum / rank, projections
//   api_precond.hip  Gram of the row sample, whitening, Ridge / preconditioner factor, its rebuild, Ridge start
//   api_solve.hip    objective, transform, the device-resident MAP solve, predictor weights, stage times
//   api_noise.hip    FunctionEstimator: noisy landmark / full conditionals, leverage, HC3 variance weights)
#pragma once
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "linalg.h"
#include "mln_internal.h"
#include "objective.h"
#include "precond_rebuild.h"
#include "solver.h"

hipError_t mln_hmalloc(void** out, size_t bytes);   // alloc.hip: page-locked host blocks, cached by size
hipError_t mln_hfree(void* p);
void fit_events_borrow(mln_ctx* ctx, std::vector<hipEvent_t>* evs);   // api_fit.hip: the context's pool of timing events
void fit_events_return(mln_ctx* ctx, std::vector<hipEvent_t>* evs);

void mln_dfree_defer(std::vector<void*>* sink);   // alloc.hip: frees of the calling thread are collected instead of performed
bool is_device_ptr(const void* p);
double now_s();

// Read-only input that may live on host or device.
struct DevIn {
  mln_ctx* ctx;
  const double* dev = nullptr;
  double* owned = nullptr;
  int init(mln_ctx* c, const double* p, size_t count) {
    ctx = c;
    if (count == 0 || !p) { dev = p; return MLN_OK; }
    if (is_device_ptr(p)) { dev = p; return MLN_OK; }
    MLN_HIP(ctx, mln_dmalloc((void**)&owned, count * sizeof(double)));
    MLN_HIP(ctx, hipMemcpyAsync(owned, p, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    dev = owned;
    return MLN_OK;
  }
  ~DevIn() {
    if (owned) { (void)hipStreamSynchronize(ctx->stream); (void)mln_dfree(owned); }
  }
};

// Output that may live on host or device.
struct DevOut {
  mln_ctx* ctx;
  double* dev = nullptr;
  double* owned = nullptr;
  double* host = nullptr;
  size_t count = 0;
  int init(mln_ctx* c, double* p, size_t n, bool copy_in = false) {
    ctx = c; count = n;
    if (n == 0) { dev = p; return MLN_OK; }
    if (is_device_ptr(p)) { dev = p; return MLN_OK; }
    host = p;
    MLN_HIP(ctx, mln_dmalloc((void**)&owned, n * sizeof(double)));
    if (copy_in) MLN_HIP(ctx, hipMemcpyAsync(owned, p, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    dev = owned;
    return MLN_OK;
  }
  int commit() {
    if (owned && count) {
      MLN_HIP(ctx, hipMemcpyAsync(host, owned, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MLN_OK;
  }
  ~DevOut() {
    if (owned) { (void)hipStreamSynchronize(ctx->stream); (void)mln_dfree(owned); }
  }
};

// ---- fit handle --------------------------------------------------------------------------------------
struct mln_fit {
  mln_ctx* ctx = nullptr;
  DevCov cov;
  int d = 0;
  int64_t n = 0, m = 0, ldl = 0, ldp = 0;
  bool full = false;
  double* L = nullptr;   // n x ldl (full GP: aliases Lp)
  double* Lp = nullptr;  // m x ldp
  TriInv tri;            // block-scaled Lp
  double *V = nullptr, *Vdr = nullptr;
  double mu = 0.0;
  // objective workspace
  int n_wg = 0;
  double *part_grad = nullptr, *part_hess = nullptr, *part_loss = nullptr;
  double *d_z = nullptr, *d_out = nullptr;  // m ; 1 + 2m
  double *h_z = nullptr, *h_out = nullptr;  // pinned
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double times[MLN_N_STAGE_TIMES] = {0};
  // preconditioner: C C^T = L^T L + I (the Ridge matrix) and C^-1, both m x ldl lower
  double *C = nullptr, *Cinv = nullptr;
  double *d_u = nullptr, *d_gu = nullptr, *d_tmp = nullptr;  // m ; m ; 1 + m
  int n_wg_cap = 0;
  // implicit ("K-space") mode: the n x m buffer holds K = cov(x, xu) itself and Lp^-T is folded
  // into the m-vectors:  L z = K (Lp^-T z),  L^T v = Lp^-1 (K^T v).  No n x m triangular solve.
  bool kspace = false;
  double* P = nullptr;    // Lp^-T C^-T  (m x ldl), so that  w = Lp^-T z = P u  for z = C^-T u
  // Round 5, implicit mode: the preconditioner is factored in w-space.  With M = s K_s^T K_s + Kj (Kj = cov(xu, xu) +
  // jitter I = Lp Lp^T) and R R^T = M, the square root C = Lp^-1 R of C C^T = I + Lp^-1 (s K_s^T K_s) Lp^-T is never
  // formed and the Gram is never whitened:  f->C holds R,  P = R^-T,  f->Cinv = C^-1 = R^-1 Lp  (both triangular).
  double* Kj = nullptr;   // cov(xu, xu) + jitter I, full symmetric (m x ldp); implicit fits only
  // MLN_FIT_DEFER_LP: f->Lp still holds Kj; it is factored together with the preconditioner's matrix (one batched chain of
  // launches, linalg.hip dev_cholesky_lower2) or on first use (fit_ensure_lp), whichever comes first
  bool lp_pending = false, lp_failed = false, tri_pending = false;
  double jitter = 0.0;
  double* d_w = nullptr;  // m
  // last vector pair (z, w = Lp^-T z) produced by the library itself (Ridge init / MAP solve): lets
  // mln_transform / mln_weights_cholesky on that same z skip the triangular solve
  std::vector<double> z_cached;
  double* d_w_cached = nullptr;
  // stacked preconditioner operators, so that one evaluation needs two row-GEMVs and no reductions:
  //   Q1 = [C^-T ; P]  (2m x ldl, implicit) or C^-T (m x ldl):   [z ; w] = Q1 u
  //   Q2 = [C^-1 | P^T] (m x 2 ldl, implicit) or C^-1:            g_u = Q2 [z ; K^T(a-1)]
  double *Q1 = nullptr, *Q2 = nullptr, *d_zw = nullptr, *d_zr = nullptr;
  // eigenvectors of L^T L (rows, ascending eigenvalue), m x ldl: Nystroem rank reduction
  double* eigU = nullptr;
  // fp32 copy of the streamed n x m buffer for the warm-up passes of the MAP solve (mixed precision)
  float* L32 = nullptr;
  double emu_excluded = 0.0;    // MELLON_AMD_EMULATE_RANKS: wall seconds spent on the OTHER ranks' column blocks (tools/emulate_rank.py)
  bool cov_bounded01 = false;   // every covariance value lies in [0, 1] (stationary kernels and their products)
  int l32_fixed = 0;     // format of that copy: 0 = fp32, 1 = 32-bit fixed point (covariances bounded by 1)
  int evals32 = 0;
  double times32 = 0.0;
  // evaluation buffers of the preconditioned objective: d_zr = [z (ld2) | r (ld2)] with the likelihood sum at
  // d_zr[ld2 + m] and the "rows above the solver's cap" flag at d_zr[ld2 + m + 1], so that one all-reduce of m + 2 values
  // covers [r ; lik ; over];  ld2 = pad16(m + 2)
  int64_t ld2 = 0;
  int* d_over = nullptr;   // device word the objective kernels set when a row lay above the cap (ObjArgs::over_flag)
  int64_t row0 = 0;     // global index of this shard's first cell (subsampling is by global index)
  // device-resident L-BFGS (solver.hip)
  SolverBuffers sv{};
  void* sv_block = nullptr;       // one allocation behind every pointer of sv
  SolverState* h_state = nullptr; // pinned mirror
  int sv_maxcor = 0;
  std::vector<hipEvent_t> evs;    // three per evaluation: before the fp32 pass, between, after the fp64 pass
  // f = L z + mu of every row at the solver's accepted point, kept by the objective passes themselves
  double* f_keep[2] = {nullptr, nullptr};
  int f_final = -1;               // which buffer holds f at z_cached (-1: none; mln_transform then streams the buffer)
  // row subsample shared by the preconditioner's Gram and the solver's first phase: cells whose GLOBAL index is a
  // multiple of precond_stride (0: no preconditioner yet; 1: all cells)
  int64_t precond_stride = 0;
  // handle whose kernel values come from the binding (mln_fit_prepare_from_K): rows received, finished
  bool from_K = false, k_finished = false;
  int64_t k_rows_done = 0;
  // the FIRST preconditioner (C, C^-1, P, Q1, Q2) while the solve runs on the rebuilt one: put back if that one fails its
  // trial (solver.h: revert_after); released when the solve ends
  double* saved_precond[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int n_revert = 0, n_rebuild_skipped = 0, n_start_halvings = 0, rank_path = 0;
  double build_seconds = 0.0;     // wall time of the first preconditioner build (Gram + factorisation): the rebuild's price
  double times_sub = 0.0, times_rebuild = 0.0, sub_pass_equiv = 0.0;
  int evals_sub = 0, n_rebuild = 0;
};

// device scratch that frees itself (after draining the stream) on every exit path
struct DevScratch {
  mln_ctx* ctx;
  double* p = nullptr;
  explicit DevScratch(mln_ctx* c) : ctx(c) {}
  hipError_t alloc(size_t bytes) { return mln_dmalloc((void**)&p, bytes); }
  ~DevScratch() {
    if (p) { (void)hipStreamSynchronize(ctx->stream); (void)mln_dfree(p); }
  }
  DevScratch(const DevScratch&) = delete;
  DevScratch& operator=(const DevScratch&) = delete;
};


int dev_allreduce(mln_ctx* ctx, double* dev, int64_t count);
int dev_bcast0(mln_ctx* ctx, double* dev, int64_t count);
int reject_distance_leaf(mln_ctx* ctx, const DevCov& dc);
int64_t pad16(int64_t m);
void fit_sample_rows(const mln_fit* f, int64_t s, int64_t* first, int64_t* rows);
void fit_free(mln_fit* f);
int fit_alloc_workspace(mln_fit* f);
int fit_prepare_impl(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n, int32_t d,
                            const double* xu, int64_t m, double jitter, const double* Lp_in, int32_t flags,
                            mln_fit* f);
ObjArgs obj_args(mln_fit* f);
void obj_account(mln_fit* f, bool f32 = false);
int fit_w_from_z(mln_fit* f, const double* z_dev, double* w_dev, const double* z_host = nullptr);
int fit_cache_pair_from_u(mln_fit* f, const double* u_dev);
int fit_enqueue_eval(mln_fit* f, const double* u_dev, double* gn_dev, bool use32, const int* gate,
                            hipEvent_t* ev, const std::vector<int64_t>* sub_strides = nullptr);
int fit_objective_u(mln_fit* f, const double* u, double* loss, double* grad_u, double* z_out,
                           bool use32 = false);
int fit_solver_alloc(mln_fit* f, int maxcor);
int gram_of(mln_ctx* ctx, const double* A, int64_t lda, int64_t rows, int64_t m, double alpha, double* G,
                   int64_t ldg, bool quantised = false);
int emulated_ranks(const mln_ctx* ctx);
int fit_ensure_kj(mln_fit* f);
int fit_ensure_lp(mln_fit* f, bool need_tri = true);   // a deferred Lp = chol(Kj) is factored now, its solve operands built (no-ops otherwise)
int fit_lp_finish(mln_fit* f, int rc_chol, double t0);   // after either route factored f->Lp: block-scaled copies, bookkeeping
// whiten = true: Lp^-1 (.) Lp^-T applied (implicit mode; a Gram whose eigenvalues are results); false: the raw K_s^T K_s
int fit_gram(mln_fit* f, double* G, int64_t ldg, int64_t row_stride, bool whiten = true);
int fit_gemvT(mln_fit* f, const double* t_dev, double* rhs_dev);
void fit_drop_precond_operators(mln_fit* f);
int fit_factor_precond(mln_fit* f);
int fit_build_precond(mln_fit* f, int64_t row_stride);
int fit_rebuild_precond(mln_fit* f, const double* f_dev, double rows_per_m, int* outcome, double cap = 1e300);   // outcome: 0 rebuilt, 1 weights too wild, 2 build failed (old one kept)
void fit_precond_saved_free(mln_fit* f);
int fit_precond_revert(mln_fit* f);
int fit_small_gemv(mln_fit* f, const double* M, int trans, const double* w, double* y);
int launch_scale_rows_cols(mln_ctx* ctx, double* A, int64_t ld, int64_t rows, int64_t cols, const double* row,
                                  const double* col);
int sparse_solve_impl(mln_ctx* ctx, const mln_kernel_desc* cov, const double* x, int64_t n_local,
                             int32_t d, const double* xu, int64_t m, const double* y, int64_t p, double mu,
                             const double* sigmas, int32_t kind, double jitter, double* W, double* Lp_out,
                             double* Cs_out);
unsigned grid_1d(int64_t count);
