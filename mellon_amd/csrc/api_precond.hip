// C ABI, part 3: Gram, whitening, Ridge / preconditioner factor and its rebuild, Ridge start (see api_internal.h).
#include "api_internal.h"
#include "mln_options.h"

// G (m x ldg, full symmetric) = alpha * A^T A for the row-major A (rows x m, leading dim lda), all-reduced.
// `quantised`: A holds covariance values in [0, 1] and the result only feeds a preconditioner -- the Gram of A rounded
// to 23 fractional bits, exact in integers on the int8 matrix cores (gram_i8.hip).
// ---- collectives that move what is there, not what the layout suggests (round 4b) --------------------------------------------
// The m x m collectives of a fit were all-reduces of full row-major matrices: the Gram (symmetric: half of it is a copy) and the
// column-split results (every rank's matrix is zero outside its own column block: a SUM with zeros of N matrices to deliver N
// blocks).  Per build 200 + 200 + 400 MB of all-reduce payload, ~4 ms on 8 xGMI-connected GPUs, twice per fit.  Now: the Gram's
// lower triangle packed (m (m + 1) / 2 doubles), summed, unpacked into both halves; the column blocks packed and ALL-GATHERED
// (each rank sends its block once and receives the others': half the traffic of the all-reduce of the same matrix, and no
// arithmetic -- the assembled matrix is the same bits on every rank by construction).
namespace {
__global__ void k_pack_lower(const double* __restrict__ G, int64_t m, int64_t ld, double* __restrict__ out) {
  const int64_t i = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j <= i) out[i * (i + 1) / 2 + j] = G[i * ld + j];
}
__global__ void k_unpack_symmetric(const double* __restrict__ in, int64_t m, int64_t ld, double* __restrict__ G) {
  const int64_t i = blockIdx.y;
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) G[i * ld + j] = (j <= i) ? in[i * (i + 1) / 2 + j] : in[j * (j + 1) / 2 + i];
}
}  // namespace

// G (m x m, lower triangle valid on every rank) <- the sum over ranks, both triangles filled
int allreduce_symmetric_lower(mln_ctx* ctx, double* G, int64_t m, int64_t ldg) {
  const int64_t cnt = m * (m + 1) / 2;
  double* packed = nullptr;
  MLN_HIP(ctx, mln_dmalloc((void**)&packed, sizeof(double) * (size_t)cnt));
  const dim3 grid((unsigned)((m + 255) / 256), (unsigned)m), block(256);
  hipLaunchKernelGGL(k_pack_lower, grid, block, 0, ctx->stream, G, m, ldg, packed);
  int rc = dev_allreduce(ctx, packed, cnt);
  if (rc == MLN_OK) {
    hipLaunchKernelGGL(k_unpack_symmetric, grid, block, 0, ctx->stream, packed, m, ldg, G);
    if (hipGetLastError() != hipSuccess) rc = MLN_ERR_HIP;
  }
  (void)hipStreamSynchronize(ctx->stream);
  (void)mln_dfree(packed);
  return rc;
}

int gram_of(mln_ctx* ctx, const double* A, int64_t lda, int64_t rows, int64_t m, double alpha, double* G,
                   int64_t ldg, bool quantised) {
  int split = quantised ? gram_i8_splits(rows, m) : (int)(rows / 8192);
  if (split < 1) split = 1;
  if (split > 16 && !quantised) split = 16;
  const size_t stride = (size_t)m * ldg;
  double* parts = nullptr;
  if (split > 1) MLN_HIP(ctx, mln_dmalloc((void**)&parts, sizeof(double) * stride * split));
  GemmArgs g{};
  g.A = A; g.lda = lda; g.B = A; g.ldb = lda;
  g.C = (split > 1) ? parts : G; g.ldc = ldg;
  g.M = m; g.N = m; g.K = rows; g.alpha = alpha; g.beta = 0.0; g.ta = 1; g.tb = 0; g.lower_only = 1;
  g.split_k = split; g.c_split_stride = (int64_t)stride;
  int rc = MLN_OK;
  if (split > 1) rc = (hipMemsetAsync(parts, 0, sizeof(double) * stride * split, ctx->stream) == hipSuccess) ? MLN_OK : MLN_ERR_HIP;
  else rc = (hipMemsetAsync(G, 0, sizeof(double) * stride, ctx->stream) == hipSuccess) ? MLN_OK : MLN_ERR_HIP;
  if (rc == MLN_OK && rows > 0)
    rc = quantised ? launch_gram_i8(ctx, A, lda, rows, m, alpha, g.C, ldg, (int64_t)stride, split) : launch_dgemm(ctx, g);
  if (rc == MLN_OK && split > 1) rc = launch_sum_partials(ctx, parts, split, (int64_t)stride, G, (int64_t)stride, 0.0);
  if (rc == MLN_OK && ctx->n_ranks > 1) rc = allreduce_symmetric_lower(ctx, G, m, ldg);      // half the payload: a Gram is symmetric
  else if (rc == MLN_OK) rc = launch_symmetrize_from_lower(ctx, G, m, ldg);
  (void)hipStreamSynchronize(ctx->stream);
  if (parts) (void)mln_dfree(parts);
  return rc;
}

// Test / measurement hook for gram_i8.hip: out (m x m) = the Gram of round(A 8355711) / 8355711^2, A rows x m with values
// in [0, 1]; ms_out (may be NULL) = milliseconds per call of digit extraction + integer GEMM + sum of the k-chunks.
extern "C" int mln_diag_gram_i8(mln_ctx* ctx, const double* A, int64_t rows, int64_t m, double* out, int32_t reps,
                                double* ms_out) {
  if (!ctx || !A || !out || rows < 1 || m < 1) return MLN_ERR_ARG;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  DevIn a;
  DevOut o;
  MLN_TRY(a.init(ctx, A, (size_t)rows * m));
  MLN_TRY(o.init(ctx, out, (size_t)m * m));
  const int split = gram_i8_splits(rows, m);
  const size_t stride = (size_t)m * m;
  double *parts = nullptr, *G = nullptr;
  MLN_HIP(ctx, mln_dmalloc((void**)&parts, sizeof(double) * stride * split));
  MLN_HIP(ctx, mln_dmalloc((void**)&G, sizeof(double) * stride));
  hipEvent_t e0, e1;
  MLN_HIP(ctx, hipEventCreate(&e0));
  MLN_HIP(ctx, hipEventCreate(&e1));
  int rc = MLN_OK;
  if (reps < 1) reps = 1;
  for (int r = 0; r <= reps && rc == MLN_OK; ++r) {   // round 0 warms up
    if (r == 1) (void)hipEventRecord(e0, ctx->stream);
    rc = (hipMemsetAsync(parts, 0, sizeof(double) * stride * split, ctx->stream) == hipSuccess) ? MLN_OK : MLN_ERR_HIP;
    if (rc == MLN_OK) rc = launch_gram_i8(ctx, a.dev, m, rows, m, 1.0, parts, m, (int64_t)stride, split);
    if (rc == MLN_OK) rc = launch_sum_partials(ctx, parts, split, (int64_t)stride, G, (int64_t)stride, 0.0);
  }
  (void)hipEventRecord(e1, ctx->stream);
  (void)hipStreamSynchronize(ctx->stream);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  if (ms_out) *ms_out = ms / reps;
  if (rc == MLN_OK) rc = launch_symmetrize_from_lower(ctx, G, m, m);
  if (rc == MLN_OK) rc = launch_copy_block(ctx, G, m, o.dev, m, m, m);
  if (rc == MLN_OK) rc = o.commit();
  (void)hipStreamSynchronize(ctx->stream);
  (void)mln_dfree(parts); (void)mln_dfree(G);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return rc;
}

// MELLON_AMD_EMULATE_RANKS=N (tools/emulate_rank.py): one process does the work of rank 0 of N -- 0 when not emulating
int emulated_ranks(const mln_ctx* ctx) {
  if (ctx->n_ranks > 1) return 0;
  if (const char* ev = std::getenv("MELLON_AMD_EMULATE_RANKS")) { const int n = std::atoi(ev); return n > 1 ? n : 0; }
  return 0;
}

// Kj = cov(xu, xu) + jitter I (full symmetric): stored by fit_prepare before the factorisation overwrites it; for handles
// that were given the factor itself, Lp Lp^T.
int fit_ensure_kj(mln_fit* f) {
  if (f->Kj) return MLN_OK;
  mln_ctx* ctx = f->ctx;
  if (!f->Lp) { mln_set_error(ctx, "this fit handle holds no Lp"); return MLN_ERR_ARG; }
  const size_t bytes = sizeof(double) * (size_t)f->m * f->ldp;
  MLN_HIP(ctx, mln_dmalloc((void**)&f->Kj, bytes));
  MLN_HIP(ctx, hipMemsetAsync(f->Kj, 0, bytes, ctx->stream));
  GemmArgs g{};
  g.A = f->Lp; g.lda = f->ldp; g.B = f->Lp; g.ldb = f->ldp; g.C = f->Kj; g.ldc = f->ldp;
  g.M = f->m; g.N = f->m; g.K = f->m; g.alpha = 1.0; g.beta = 0.0; g.ta = 0; g.tb = 1; g.lower_only = 1;
  MLN_TRY(launch_dgemm(ctx, g));
  return launch_symmetrize_from_lower(ctx, f->Kj, f->m, f->ldp);
}

// G ~ L^T L from every `row_stride`-th cell of this rank (scaled by row_stride), all-reduced.
// Implicit mode: the n x m buffer holds K, so the Gram of its rows is S = s K_s^T K_s.
//   whiten = true   G = Lp^-1 S Lp^-T = L_s^T L_s by two backward-stable block solves: the callers whose Gram is a RESULT
//                   (spectrum / rank of L, the noisy conditionals of the function estimator);
//   whiten = false  G = S itself: the preconditioner, which is factored in w-space (fit_build_precond) and never whitened.
int fit_gram(mln_fit* f, double* G, int64_t ldg, int64_t row_stride, bool whiten) {
  mln_ctx* ctx = f->ctx;
  if (row_stride < 1) row_stride = 1;
  // cells whose GLOBAL index is a multiple of row_stride: the sample -- and with it the preconditioner and the
  // iteration path -- does not depend on how the cells are sharded (up to the order of the all-reduce sum)
  const int64_t first = (row_stride - f->row0 % row_stride) % row_stride;
  const int64_t rows = (f->n > first) ? (f->n - first + row_stride - 1) / row_stride : 0;
  const double* Ls = f->L + first * f->ldl;
  if (!f->kspace) return gram_of(ctx, Ls, f->ldl * row_stride, rows, f->m, (double)row_stride, G, ldg);
  // bounded covariances: 23-bit integer Gram on the int8 matrix cores (the preconditioner needs ~20 bits: gram_i8.hip)
  // ... and only where the caller asked for a SAMPLED Gram (row_stride > 1: a preconditioner or a diagnostic by
  // construction); row_stride == 1 is the reference's exact Ridge matrix / the Gram whose eigenvalues are results
  bool quant = f->cov_bounded01 && f->m >= 256 && row_stride > 1;
  if (const char* ev = mln_experiment("MELLON_AMD_GRAM_I8")) quant = quant && std::atoi(ev) != 0;
  int rc = gram_of(ctx, Ls, f->ldl * row_stride, rows, f->m, (double)row_stride, G, ldg, quant);   // all-reduced
  if (rc != MLN_OK || !whiten) return rc;
  MLN_TRY(fit_ensure_lp(f));
  double* T = nullptr;
  {
    hipError_t e = mln_dmalloc((void**)&T, sizeof(double) * (size_t)f->m * ldg);
    if (e == hipSuccess) e = hipMemsetAsync(T, 0, sizeof(double) * (size_t)f->m * ldg, ctx->stream);
    if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "alloc Gram temp", __FILE__, __LINE__);
  }
  if (rc == MLN_OK) rc = triinv_solve_left(ctx, f->tri, G, f->m, ldg);          // Lp^-1 S
  if (rc == MLN_OK) rc = launch_transpose(ctx, G, ldg, T, ldg, f->m);           // (Lp^-1 S)^T
  if (rc == MLN_OK) rc = triinv_solve_left(ctx, f->tri, T, f->m, ldg);          // Lp^-1 S Lp^-T (symmetric)
  if (rc == MLN_OK) rc = (hipMemcpyAsync(G, T, sizeof(double) * (size_t)f->m * ldg, hipMemcpyDeviceToDevice,
                                         ctx->stream) == hipSuccess) ? MLN_OK : MLN_ERR_HIP;
  (void)hipStreamSynchronize(ctx->stream);
  if (T) (void)mln_dfree(T);
  return rc;
}

// rhs (m) = L^T t over this rank's rows, all-reduced; t is a device vector of length n
int fit_gemvT(mln_fit* f, const double* t_dev, double* rhs_dev) {
  mln_ctx* ctx = f->ctx;
  ObjArgs a = obj_args(f);
  a.weights = t_dev;
  a.part_loss = nullptr;
  MLN_TRY(launch_objective(ctx, a));
  MLN_TRY(launch_reduce_obj(ctx, a, f->d_out));
  MLN_TRY(dev_allreduce(ctx, f->d_out, 1 + f->m));
  if (f->kspace) { MLN_TRY(fit_ensure_lp(f)); MLN_TRY(triinv_solve_left(ctx, f->tri, f->d_out + 1, 1, 1)); }   // Lp^-1 (K^T t)
  MLN_HIP(ctx, hipMemcpyAsync(rhs_dev, f->d_out + 1, sizeof(double) * f->m, hipMemcpyDeviceToDevice, ctx->stream));
  return MLN_OK;
}

// C C^T = L^T L + I and C^-1 (explicit, lower): the Ridge matrix of parameters.py:895-896 doubles as
// the preconditioner of the MAP solve, because the MAP Hessian I + L^T diag(e^{f+V}) L equals it
// wherever e^{f+V} = 1 (i.e. where f matches the nearest-neighbour estimate the Ridge regresses on).
// With row_stride > 1 the Gram is estimated from every row_stride-th cell: any SPD matrix is a valid
// preconditioner / initial guess for a strictly convex problem, and ~8 m rows already give the same
// iteration count as all n (measured), at 1/row_stride of the n m^2 flops.
void fit_drop_precond_operators(mln_fit* f) {
  (void)hipStreamSynchronize(f->ctx->stream);
  void* ptrs[] = {f->Cinv, f->P, f->Q1, f->Q2};
  for (void* p : ptrs) if (p) (void)mln_dfree(p);
  f->Cinv = nullptr; f->P = nullptr; f->Q1 = nullptr; f->Q2 = nullptr;
}

// f->C holds the matrix to factor -- explicit mode: the Gram L_s^T L_s (the prior's identity is added here); implicit
// mode: M = s K_s^T K_s + Kj, the same Hessian at a = 1 in w-space (w = Lp^-T z), never whitened.
// Explicit mode: C C^T = that, C^-1, and the stacked per-evaluation operators Q1 = C^-T, Q2 = [C^-1 | C^-1].
// Implicit mode (round 5): R R^T = M.  C = Lp^-1 R satisfies C C^T = I + Lp^-1 (s K_s^T K_s) Lp^-T -- the matrix rounds
// 2-4 obtained by whitening the Gram with an explicit Lp^-1 (1.5 ms) and two m^3 GEMMs (3.7 ms) and then factored.  C is
// never formed.  With z = C^-T u the evaluation needs  w = Lp^-T z = R^-T u,  the prior 1/2 |z|^2 = 1/2 w^T Kj w  and
// g_u = C^-1 z + R^-1 K^T (a - 1) = R^-1 (Kj w + K^T (a - 1)):  P = R^-T, R^-1 and Kj itself -- no product with Lp at all,
// so the factor Lp is not even needed before the solve ends (fit_prepare may defer it: it then shares THIS chain of
// launches, dev_cholesky_lower2).  M -- a sum of a positive semi-definite integer Gram and Kj -- cannot lose positive
// definiteness to the quantisation of its rows the way the whitened matrix did on heavy-tailed data (round 4).
//   f->C = R,  f->Cinv = R^-1 (lower),  f->P = R^-T (upper);  Q1, Q2 stay empty.
int fit_factor_precond(mln_fit* f) {
  mln_ctx* ctx = f->ctx;
  const int64_t m = f->m, ldg = f->ldl;
  const size_t bytes = sizeof(double) * (size_t)m * ldg;
  int rc = f->kspace ? MLN_OK : launch_add_diag(ctx, f->C, m, ldg, 1.0);  // Ridge alpha = 1 / the prior's Hessian
  if (rc == MLN_OK && f->kspace && f->lp_pending && !f->lp_failed && f->ldp == ldg) {
    // both factorisations in one chain: its 40 dependent block steps are latency, the second matrix only adds flops
    const double t0 = now_s();
    int bad = 0;
    rc = dev_cholesky_lower2(ctx, f->C, f->Lp, m, ldg, &bad);
    const int rc_lp = (rc == MLN_ERR_NOT_PD && !(bad & 2)) ? MLN_OK : rc;   // (bit 1 clear: only the preconditioner's matrix failed, Lp is fine)
    if (rc == MLN_OK || rc == MLN_ERR_NOT_PD) {
      const std::string msg = ctx->err;
      const int rl = fit_lp_finish(f, rc_lp, t0);
      if (rl != MLN_OK) return rl;                       // "cov(xu, xu) ...": the landmarks' own matrix is the failure
      if (rc != MLN_OK) mln_set_error(ctx, msg);
    }
  } else if (rc == MLN_OK) {
    rc = dev_cholesky_lower(ctx, f->C, m, ldg);
  }
  TriInv t;
  if (rc == MLN_OK) rc = triinv_build(ctx, f->C, m, ldg, true, false, &t);
  double* inv = nullptr;                             // the factor's explicit inverse
  auto zeroed = [&](double** p, const char* what) {
    hipError_t e = mln_dmalloc((void**)p, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(*p, 0, bytes, ctx->stream);
    if (e != hipSuccess) rc = mln_hip_fail(ctx, e, what, __FILE__, __LINE__);
  };
  if (rc == MLN_OK) zeroed(&inv, "alloc factor inverse");
  if (rc == MLN_OK) rc = launch_add_diag(ctx, inv, m, ldg, 1.0);
  if (rc == MLN_OK) rc = triinv_solve_left(ctx, t, inv, m, ldg, true);   // R^-1 I (lower triangular right-hand side)
  if (rc == MLN_OK && f->kspace) {
    zeroed(&f->P, "alloc P");
    if (rc == MLN_OK) rc = launch_transpose(ctx, inv, ldg, f->P, ldg, m);        // P = R^-T
  } else if (rc == MLN_OK) {   // stacked operators for the per-evaluation row-GEMVs of the explicit factor
    const int64_t ld = ldg;
    const size_t blk = (size_t)m * ld;
    hipError_t e = mln_dmalloc((void**)&f->Q1, sizeof(double) * blk);
    if (e == hipSuccess) e = mln_dmalloc((void**)&f->Q2, sizeof(double) * blk * 2);
    if (e == hipSuccess) e = hipMemsetAsync(f->Q1, 0, sizeof(double) * blk, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(f->Q2, 0, sizeof(double) * blk * 2, ctx->stream);
    if (e != hipSuccess) rc = mln_hip_fail(ctx, e, "alloc stacked operators", __FILE__, __LINE__);
    if (rc == MLN_OK) rc = launch_transpose(ctx, inv, ld, f->Q1, ld, m);                      // C^-T
    if (rc == MLN_OK) rc = launch_copy_block(ctx, inv, ld, f->Q2, ld * 2, m, ld);             // C^-1
    // g_u = C^-1 (z + L^T(a-1)) = [C^-1 | C^-1] [z ; r] -- a two-segment product
    if (rc == MLN_OK) rc = launch_copy_block(ctx, inv, ld, f->Q2 + ld, ld * 2, m, ld);
  }
  (void)hipStreamSynchronize(ctx->stream);
  triinv_free(&t);
  if (rc == MLN_OK) f->Cinv = inv; else if (inv) (void)mln_dfree(inv);
  return rc;
}

int fit_build_precond(mln_fit* f, int64_t row_stride) {
  if (f->Cinv) return MLN_OK;
  mln_ctx* ctx = f->ctx;
  const int64_t m = f->m, ldg = f->ldl;
  const size_t bytes = sizeof(double) * (size_t)m * ldg;
  double t0 = now_s();
  MLN_HIP(ctx, mln_dmalloc((void**)&f->C, bytes));
  int rc = fit_gram(f, f->C, ldg, row_stride, false);
  if (rc == MLN_OK && f->kspace) rc = fit_ensure_kj(f);
  if (rc == MLN_OK && f->kspace) rc = launch_axpby(ctx, m * ldg, 1.0, f->Kj, 1.0, f->C);      // M = s K_s^T K_s + Kj
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  f->times[3] += now_s() - t0;
  double t1 = now_s();
  if (rc == MLN_OK) rc = fit_factor_precond(f);
  f->times[4] += now_s() - t1;
  if (rc == MLN_OK) { f->precond_stride = row_stride < 1 ? 1 : row_stride; f->build_seconds = now_s() - t0; }
  return rc;
}

// The solver's SECOND preconditioner (precond_rebuild.hip): C C^T = I + sum_i a_i L_i L_i^T estimated from an importance
// sample of ~rows_per_m * m cells at the point whose rows' f = L z + mu is `f_dev`; replaces C, C^-1, P, Q1, Q2.
void fit_precond_saved_free(mln_fit* f) {
  bool any = false;
  for (double* p : f->saved_precond) any = any || p;
  if (!any) return;
  (void)hipStreamSynchronize(f->ctx->stream);
  for (double*& p : f->saved_precond) { if (p) (void)mln_dfree(p); p = nullptr; }
}

// the saved (first) preconditioner becomes the current one again; the current one is released
int fit_precond_revert(mln_fit* f) {
  if (!f->saved_precond[0] || !f->saved_precond[1]) return MLN_ERR_ARG;
  fit_drop_precond_operators(f);
  if (f->C) { (void)mln_dfree(f->C); f->C = nullptr; }
  f->C = f->saved_precond[0]; f->Cinv = f->saved_precond[1]; f->P = f->saved_precond[2];
  f->Q1 = f->saved_precond[3]; f->Q2 = f->saved_precond[4];
  for (double*& p : f->saved_precond) p = nullptr;
  return MLN_OK;
}

// Round 4 (tools/robustness_sweep_large.py): the rebuild may decline or fail, and the solve then goes on with the
// preconditioner it has.  outcome 1: the sample's weights span more than 1e5 (w_max c: heaviest weight over the floor
// 1 / c) -- the pause came on a plateau far from the optimum, the scaled rows' 23-bit digits would hold only the heaviest
// cells, and the factor built from them stalled the solve for thousands of passes on tree-shaped data.  outcome 2: the
// Gram's whitening lost positive definiteness (heavy-tailed data, w_max ~ 1e8: "Covariance not positively definite").
// On success the FIRST preconditioner is kept in f->saved_precond until the solve ends (solver.h: revert_after).
int fit_rebuild_precond(mln_fit* f, const double* f_dev, double rows_per_m, int* outcome, double cap) {
  mln_ctx* ctx = f->ctx;
  const int64_t m = f->m, ldg = f->ldl;
  *outcome = 0;
  RebuildSelection sel{};
  double target = rows_per_m * (double)m;
  // (tools/emulate_rank.py: one process stands for rank 0 of N and sees only its shard -- "global" sums are local there, and
  //  the importance sample would come out N times this rank's real share: 30 000 rows instead of 3 750 at 8 ranks, a Gram
  //  eight times too expensive.  The emulation asks for the share.)
  if (const int n_emu = emulated_ranks(ctx)) target /= (double)n_emu;
  const bool tr_on = std::getenv("MELLON_AMD_TRACE") != nullptr;
  double tt[6] = {0, 0, 0, 0, 0, 0};
  auto lap = [&](int i, double& t0) { if (tr_on) { (void)hipStreamSynchronize(ctx->stream); const double t1 = now_s(); tt[i] += t1 - t0; t0 = t1; } };
  double tl = now_s();
  // (Clipping the weights to a range of 1e3 .. 1e5 instead of declining was tried -- tools/clip_sweep.py,
  //  profiles/r04_clip_sweep.txt: the clipped preconditioner failed its trial on the tree and helped nowhere.)
  MLN_TRY(rebuild_select_rows(ctx, f_dev, f->V, f->n, f->row0, target, 0x6d656c6c6f6eull, &sel, cap));
  lap(0, tl);
  if (tr_on)
    fprintf(stderr, "[trace] rebuild: %lld of %lld local rows kept (target %.0f global), c = %.4g, 1/c = %.4g, w_max = %.4g, sum a = %.6g\n",
            (long long)sel.rows, (long long)f->n, target, sel.c, 1.0 / sel.c, sel.w_max, sel.sum_a);
  double range_cap = 1e7;      // (round 5: was 1e5 -- 11 of the 23 bits are left to the lightest rows at 1e7; the w-space factor cannot lose
                               //  positive definiteness to their rounding)
  if (const char* ev = mln_experiment("MELLON_AMD_REBUILD_RANGE")) range_cap = std::atof(ev);
  if (!(sel.w_max * sel.c <= range_cap) || !std::isfinite(sel.sum_a)) {   // (global quantities: the same decision on every rank)
    rebuild_selection_free(ctx, &sel);
    *outcome = 1;
    return MLN_OK;
  }
  double* R = nullptr;
  int rc = MLN_OK;
  const int64_t rr = sel.rows > 0 ? sel.rows : 1;
  if (mln_dmalloc((void**)&R, sizeof(double) * (size_t)rr * f->ldl) != hipSuccess) rc = MLN_ERR_HIP;
  if (rc == MLN_OK) rc = launch_gather_scale_rows(ctx, f->L, f->ldl, sel.idx, sel.scale, sel.rows, R);
  // the current preconditioner steps aside (kept: fallback now, revert later); the new one is built in fresh buffers
  double* old_set[5] = {f->C, f->Cinv, f->P, f->Q1, f->Q2};
  f->C = nullptr; f->Cinv = nullptr; f->P = nullptr; f->Q1 = nullptr; f->Q2 = nullptr;
  if (rc == MLN_OK && mln_dmalloc((void**)&f->C, sizeof(double) * (size_t)m * ldg) != hipSuccess) rc = MLN_ERR_HIP;
  lap(1, tl);
  if (rc == MLN_OK) {
    // scaled covariances stay in [0, 1]: the integer Gram applies where it did for the first preconditioner
    bool quant = f->kspace && f->cov_bounded01 && m >= 256;
    if (const char* ev = mln_experiment("MELLON_AMD_GRAM_I8")) quant = quant && std::atoi(ev) != 0;
    rc = gram_of(ctx, R, f->ldl, sel.rows, m, sel.w_max, f->C, ldg, quant);                   // all-reduced
    // (The integer Gram is that of the rows ROUNDED to 1 / 8355711; the rounding's own Gram, rows * var * I times w_max
    //  and the whitening's |Lp^-1|^2, is an O(0.1) multiple of K_uu^-1.  Subtracting its expectation was tried: no change
    //  in the pass count at w_max ~ 5e3, and at w_max ~ 1e5 the subtraction itself made the matrix indefinite.)
  }
  (void)hipStreamSynchronize(ctx->stream);
  if (R) (void)mln_dfree(R);
  rebuild_selection_free(ctx, &sel);
  lap(2, tl);
  if (rc == MLN_OK && f->kspace) rc = fit_ensure_kj(f);
  if (rc == MLN_OK && f->kspace) rc = launch_axpby(ctx, m * ldg, 1.0, f->Kj, 1.0, f->C);       // M' = sum_i a_i K_i K_i^T + Kj
  lap(3, tl);
  if (rc == MLN_OK) rc = fit_factor_precond(f);
  lap(4, tl);
  if (tr_on) fprintf(stderr, "[trace] rebuild ms: select %.2f, gather %.2f, gram %.2f, + Kj %.2f, factor+inverses+stacks %.2f (rc %d)\n",
                     1e3 * tt[0], 1e3 * tt[1], 1e3 * tt[2], 1e3 * tt[3], 1e3 * tt[4], rc);
  if (rc == MLN_ERR_NOT_PD) {
    // (the factorisation's verdict is a function of all-reduced numbers: every rank lands here together)
    fit_drop_precond_operators(f);
    if (f->C) { (void)hipStreamSynchronize(ctx->stream); (void)mln_dfree(f->C); }
    f->C = old_set[0]; f->Cinv = old_set[1]; f->P = old_set[2]; f->Q1 = old_set[3]; f->Q2 = old_set[4];
    *outcome = 2;
    return MLN_OK;
  }
  if (rc != MLN_OK) {                        // a real failure: leave the handle consistent (old set back), report it
    fit_drop_precond_operators(f);
    if (f->C) { (void)hipStreamSynchronize(ctx->stream); (void)mln_dfree(f->C); }
    f->C = old_set[0]; f->Cinv = old_set[1]; f->P = old_set[2]; f->Q1 = old_set[3]; f->Q2 = old_set[4];
    return rc;
  }
  fit_precond_saved_free(f);
  for (int i = 0; i < 5; ++i) f->saved_precond[i] = old_set[i];
  return MLN_OK;
}

// y (m) = M^T w  (trans = 1)  or  M w  (trans = 0) for an m x ldl matrix M, via the streaming kernels
// of objective.hip (GEMV-T mode / f-only mode); all pointers on the device.
int fit_small_gemv(mln_fit* f, const double* M, int trans, const double* w, double* y) {
  mln_ctx* ctx = f->ctx;
  ObjArgs a{};
  a.L = M; a.ldl = f->ldl; a.n = f->m; a.m = f->m; a.mu = 0.0;
  a.part_grad = f->part_grad; a.part_hess = nullptr; a.part_loss = nullptr;
  a.m_pad = f->ldl;
  int64_t steps = (f->m + 1) / 2;
  a.n_wg = (int)((steps < f->n_wg_cap) ? (steps > 0 ? steps : 1) : f->n_wg_cap);
  if (trans) {
    a.weights = w;
    MLN_TRY(launch_objective(ctx, a));
    MLN_TRY(launch_reduce_obj(ctx, a, f->d_tmp));
    MLN_HIP(ctx, hipMemcpyAsync(y, f->d_tmp + 1, sizeof(double) * f->m, hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    a.z = w;
    a.f_out = y;
    MLN_TRY(launch_objective(ctx, a));
  }
  return MLN_OK;
}

extern "C" int mln_fit_set_row_offset(mln_fit* f, int64_t global_row0) {
  if (!f || global_row0 < 0) return MLN_ERR_ARG;
  f->row0 = global_row0;
  return MLN_OK;
}

extern "C" int mln_precond_build(mln_fit* f, int64_t row_stride) {
  if (!f) return MLN_ERR_ARG;
  MLN_HIP(f->ctx, hipSetDevice(f->ctx->device));
  if (row_stride < 1) row_stride = 1;
  if (f->Cinv && f->precond_stride != row_stride) {
    // an explicit request for a DIFFERENT sample (e.g. the reference's exact Ridge, stride 1, after a sampled
    // preconditioner had been built): drop the factor and build the one asked for
    fit_drop_precond_operators(f);
    if (f->C) { (void)mln_dfree(f->C); f->C = nullptr; }
    f->precond_stride = 0;
  }
  return fit_build_precond(f, row_stride);
}

extern "C" int mln_ridge_init(mln_fit* f, const double* target, double* z0) {
  if (!f || !z0 || (f->n > 0 && !target)) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  MLN_TRY(fit_build_precond(f, 1));   // exact Ridge unless a (subsampled) factor was built before
  double t0 = now_s();
  DevIn dt;
  MLN_TRY(dt.init(ctx, target, (size_t)f->n));
  // z0 = (L^T L + I)^-1 L^T t = C^-T C^-1 (L^T t);  implicit mode: C^-1 L^T t = P^T (K^T t)
  // With a sampled Gram (stride s >= 11) the right-hand side is taken over the SAME cells, s L_s^T t_s: z0 is then the
  // exact Ridge solution of the subsample -- the problem the solver's first phase works on -- and costs 1/s of a pass.
  // (Not beyond 8192 landmarks: the segmented pass, launch_objective_wide, has no row map; there the right-hand side
  //  runs over all cells against the sampled Gram -- a valid start for a solve that the host's L-BFGS-B drives anyway.)
  int64_t rs = (f->precond_stride >= 11 && f->m <= objective_max_m_one_pass()) ? f->precond_stride : 1;
  if (const char* ev = mln_experiment("MELLON_AMD_SUBSAMPLE")) { if (std::atoi(ev) == 0) rs = 1; }
  ObjArgs a = obj_args(f);
  a.weights = dt.dev;
  a.part_loss = nullptr;
  if (rs > 1) {
    int64_t first = 0, rows = 0;
    fit_sample_rows(f, rs, &first, &rows);
    a.n = rows; a.row_first = first; a.row_stride = rs; a.out_scale = (double)rs;
  }
  if (f->kspace) {
    a.L32 = f->L32;   // the Ridge solution only seeds the solve: its right-hand side may come from the 32-bit copy
    a.l32_fixed = f->l32_fixed;
    MLN_TRY(launch_objective(ctx, a));
    MLN_TRY(launch_reduce_obj(ctx, a, f->d_out));
    MLN_TRY(dev_allreduce(ctx, f->d_out, 1 + f->m));
    // u0 = R^-1 (s K_s^T t) ;  w0 = R^-T u0 ;  z0 = Lp^T w0
    MLN_TRY(fit_small_gemv(f, f->Cinv, 0, f->d_out + 1, f->d_gu));
    MLN_TRY(fit_small_gemv(f, f->P, 0, f->d_gu, f->d_w));
    MLN_TRY(fit_ensure_lp(f, false));
    MLN_TRY(fit_small_gemv(f, f->Lp, 1, f->d_w, f->d_z));
  } else {
    MLN_TRY(launch_objective(ctx, a));
    MLN_TRY(launch_reduce_obj(ctx, a, f->d_out));
    MLN_TRY(dev_allreduce(ctx, f->d_out, 1 + f->m));
    MLN_TRY(fit_small_gemv(f, f->Cinv, 0, f->d_out + 1, f->d_gu));
  }
  if (!f->kspace) MLN_TRY(fit_small_gemv(f, f->Cinv, 1, f->d_gu, f->d_z));       // z0 = C^-T (.)   [d_gu plays the role of u0]
  MLN_TRY(fit_cache_pair_from_u(f, f->d_gu));
  MLN_HIP(ctx, hipMemcpyAsync(z0, f->d_z, sizeof(double) * f->m, hipMemcpyDefault, ctx->stream));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  f->times[4] += now_s() - t0;
  return MLN_OK;
}

// u <-> z of the preconditioned variable  z = C^-T u
extern "C" int mln_precond_apply(mln_fit* f, int32_t mode, const double* in, double* out) {
  if (!f || !in || !out) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  MLN_TRY(fit_build_precond(f, 1));
  MLN_HIP(ctx, hipMemcpyAsync(f->d_u, in, sizeof(double) * f->m, hipMemcpyDefault, ctx->stream));
  if (f->kspace && mode >= 0 && mode <= 2) {
    // C = Lp^-1 R (never formed):  u = C^T z = R^T (Lp^-T z) ;  z = C^-T u = Lp^T (R^-T u) ;  g_u = C^-1 g_z = R^-1 (Lp g_z)
    MLN_TRY(fit_ensure_lp(f));
    if (mode == 0) {
      MLN_TRY(fit_w_from_z(f, f->d_u, f->d_w, is_device_ptr(in) ? nullptr : in));
      MLN_TRY(fit_small_gemv(f, f->C, 1, f->d_w, f->d_gu));
    } else if (mode == 1) {
      MLN_TRY(fit_small_gemv(f, f->P, 0, f->d_u, f->d_w));
      MLN_TRY(fit_small_gemv(f, f->Lp, 1, f->d_w, f->d_gu));
    } else {
      MLN_TRY(fit_small_gemv(f, f->Lp, 0, f->d_u, f->d_w));
      MLN_TRY(fit_small_gemv(f, f->Cinv, 0, f->d_w, f->d_gu));
    }
  }
  else if (mode == 0) MLN_TRY(fit_small_gemv(f, f->C, 1, f->d_u, f->d_gu));     // u = C^T z
  else if (mode == 1) MLN_TRY(fit_small_gemv(f, f->Cinv, 1, f->d_u, f->d_gu));  // z = C^-T u
  else if (mode == 2) MLN_TRY(fit_small_gemv(f, f->Cinv, 0, f->d_u, f->d_gu));  // g_u = C^-1 g_z
  else { mln_set_error(ctx, "mln_precond_apply: unknown mode"); return MLN_ERR_ARG; }
  MLN_HIP(ctx, hipMemcpyAsync(out, f->d_gu, sizeof(double) * f->m, hipMemcpyDefault, ctx->stream));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MLN_OK;
}

