// C ABI, part 4: objective, transform, MAP solve, predictor weights (see api_internal.h).
#include "api_internal.h"
#include "mln_options.h"


void obj_account(mln_fit* f, bool f32) {
  float ms = 0.f;
  const bool ok = hipEventElapsedTime(&ms, f->ev0, f->ev1) == hipSuccess;
  if (f32) {   // fp32 warm-up passes are accounted separately: the roofline figure is the fp64 kernel's
    if (ok) f->times32 += 1e-3 * ms;
    f->evals32 += 1;
    return;
  }
  if (ok) f->times[5] += 1e-3 * ms;
  f->times[6] += 1.0;
  f->times[7] = (double)f->n * (double)f->ldl * 8.0;
}

// In implicit mode the streamed matrix is K and the kernel's vector is w = Lp^-T z (device, m).
// `z_host` (may be NULL) is the caller's host copy of z, used to recognise the cached pair.
int fit_w_from_z(mln_fit* f, const double* z_dev, double* w_dev, const double* z_host) {
  mln_ctx* ctx = f->ctx;
  if (z_host && f->z_cached.size() == (size_t)f->m &&
      std::memcmp(z_host, f->z_cached.data(), sizeof(double) * f->m) == 0) {
    MLN_HIP(ctx, hipMemcpyAsync(w_dev, f->d_w_cached, sizeof(double) * f->m, hipMemcpyDeviceToDevice, ctx->stream));
    return MLN_OK;
  }
  MLN_TRY(fit_ensure_lp(f));
  MLN_HIP(ctx, hipMemcpyAsync(w_dev, z_dev, sizeof(double) * f->m, hipMemcpyDeviceToDevice, ctx->stream));
  return triinv_solve_left_T(ctx, f->tri, w_dev, 1, 1);
}

// remember (z, w) computed from the preconditioned variable: z = C^-T u (already in d_z), w = P u
int fit_cache_pair_from_u(mln_fit* f, const double* u_dev) {
  mln_ctx* ctx = f->ctx;
  f->z_cached.assign((size_t)f->m, 0.0);
  MLN_HIP(ctx, hipMemcpyAsync(f->z_cached.data(), f->d_z, sizeof(double) * f->m, hipMemcpyDeviceToHost, ctx->stream));
  if (f->kspace) MLN_TRY(fit_small_gemv(f, f->P, 0, u_dev, f->d_w_cached));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MLN_OK;
}

extern "C" int mln_objective(mln_fit* f, const double* z, double* loss, double* grad, double* hess_diag) {
  if (!f || !z || !loss || !grad) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  if (!f->V) { mln_set_error(ctx, "mln_fit_set_likelihood has not been called"); return MLN_ERR_ARG; }
  if (hess_diag && f->kspace) {
    mln_set_error(ctx, "the Hessian diagonal needs the explicit factor L: prepare the fit without MLN_FIT_IMPLICIT");
    return MLN_ERR_UNSUPPORTED;
  }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  MLN_TRY(fit_ensure_lp(f));
  const int64_t m = f->m;
  MLN_HIP(ctx, hipMemcpyAsync(f->d_z, z, sizeof(double) * m, hipMemcpyDefault, ctx->stream));
  MLN_HIP(ctx, hipMemcpyAsync(f->h_z, f->d_z, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
  ObjArgs a = obj_args(f);
  if (f->kspace) {
    MLN_TRY(fit_w_from_z(f, f->d_z, f->d_w));
    a.z = f->d_w;
  }
  if (hess_diag) a.part_hess = f->part_hess;
  const int64_t nout = 1 + m + (hess_diag ? m : 0);
  MLN_HIP(ctx, hipEventRecord(f->ev0, ctx->stream));
  MLN_TRY(launch_objective(ctx, a));
  MLN_HIP(ctx, hipEventRecord(f->ev1, ctx->stream));
  MLN_TRY(launch_reduce_obj(ctx, a, f->d_out));
  MLN_TRY(dev_allreduce(ctx, f->d_out, nout));
  if (f->kspace) MLN_TRY(triinv_solve_left(ctx, f->tri, f->d_out + 1, 1, 1));   // L^T v = Lp^-1 (K^T v)
  MLN_HIP(ctx, hipMemcpyAsync(f->h_out, f->d_out, sizeof(double) * nout, hipMemcpyDeviceToHost, ctx->stream));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  obj_account(f);
  // prior terms, added once (inference.py:45-46): 1/2 |z|^2 + (k/2) log 2 pi ; d/dz = z ; d2/dz2 = 1
  double zz = 0.0;
  for (int64_t j = 0; j < m; ++j) zz += f->h_z[j] * f->h_z[j];
  *loss = f->h_out[0] + 0.5 * zz + 0.5 * (double)m * std::log(2.0 * M_PI);
  std::vector<double> tmp;
  double* gh = grad;
  if (is_device_ptr(grad)) { tmp.resize(m); gh = tmp.data(); }
  for (int64_t j = 0; j < m; ++j) gh[j] = f->h_out[1 + j] + f->h_z[j];
  if (gh != grad) MLN_HIP(ctx, hipMemcpy(grad, gh, sizeof(double) * m, hipMemcpyHostToDevice));
  if (hess_diag) {
    std::vector<double> th;
    double* hh = hess_diag;
    if (is_device_ptr(hess_diag)) { th.resize(m); hh = th.data(); }
    for (int64_t j = 0; j < m; ++j) hh[j] = f->h_out[1 + m + j] + 1.0;
    if (hh != hess_diag) MLN_HIP(ctx, hipMemcpy(hess_diag, hh, sizeof(double) * m, hipMemcpyHostToDevice));
  }
  return MLN_OK;
}

extern "C" int mln_transform(mln_fit* f, const double* z, double mu, double* f_out) {
  if (!f || !z || (f->n > 0 && !f_out)) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  if (f->n == 0) return MLN_OK;
  MLN_HIP(ctx, hipMemcpyAsync(f->d_z, z, sizeof(double) * f->m, hipMemcpyDefault, ctx->stream));
  if (f->f_final >= 0 && mu == f->mu && !is_device_ptr(z) && f->z_cached.size() == (size_t)f->m &&
      std::memcmp(z, f->z_cached.data(), sizeof(double) * f->m) == 0) {
    // the last accepted pass of the MAP solve stored exactly this vector (same kernel, same operands)
    MLN_HIP(ctx, hipMemcpyAsync(f_out, f->f_keep[f->f_final], sizeof(double) * (size_t)f->n, hipMemcpyDefault, ctx->stream));
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MLN_OK;
  }
  DevOut o;
  MLN_TRY(o.init(ctx, f_out, (size_t)f->n));
  ObjArgs a = obj_args(f);
  if (f->kspace) {
    MLN_TRY(fit_w_from_z(f, f->d_z, f->d_w, is_device_ptr(z) ? nullptr : z));
    a.z = f->d_w;
  }
  a.f_out = o.dev;
  a.mu = mu;
  MLN_TRY(launch_objective(ctx, a));
  return o.commit();
}

// One evaluation of the preconditioned objective at the device vector `u`, enqueued without any host wait:
//   explicit mode: z = C^-T u  ->  one pass over the n x m buffer (f = L z + mu)  ->  fixed-order reduction  ->  all-reduce
//                  of [r ; lik]  ->  g_u = C^-1 (z + r)      (z -> d_zr, r -> d_zr + ld2, lik -> d_zr[ld2 + m], g_u -> gn)
//   implicit mode: w = R^-T u (d_w),  q = Kj w (d_zr)  ->  the pass (f = K w + mu)  ->  reduction, all-reduce
//                  ->  g_u = R^-1 (q + r);  prior = 1/2 w . q   (api_precond.hip fit_factor_precond)
// gate == nullptr: `use32` picks the streamed copy.  gate != nullptr (device-resident solver): both objective kernels
// are launched and the one the solver's state does not select returns at once; everything is a no-op after DONE.
// ev (optional): three events -- before the fp32 pass, between the two, after the fp64 pass.
int fit_enqueue_eval(mln_fit* f, const double* u_dev, double* gn_dev, bool use32, const int* gate,
                            hipEvent_t* ev, const std::vector<int64_t>* sub_strides) {
  mln_ctx* ctx = f->ctx;
  const int64_t m = f->m, ld = f->ldl, ld2 = f->ld2;
  if (f->kspace) {
    // w = R^-T u (upper triangular rows), then q = Kj w (full rows; the solver's prior is 1/2 w . q)
    GemvTri g1{f->P, ld, m, u_dev, f->d_w, nullptr, 1, m, m, 0, 0, gate};
    MLN_TRY(launch_gemv_tri(ctx, g1));
    GemvTri gk{f->Kj, f->ldp, m, f->d_w, f->d_zr, nullptr, 2, m, m, 0, 0, gate};
    MLN_TRY(launch_gemv_tri(ctx, gk));
  } else {
    GemvTri g1{f->Q1, ld, m, u_dev, f->d_zr, nullptr, 1, m, m, 0, 0, gate};
    MLN_TRY(launch_gemv_tri(ctx, g1));                                 // z = C^-T u: upper triangular rows
  }
  ObjArgs a = obj_args(f);
  a.z = f->kspace ? f->d_w : f->d_zr;
  a.gate = gate;
  if (gate && f->f_keep[0] && objective_can_keep_f(f->n, f->n_wg)) { a.f_keep[0] = f->f_keep[0]; a.f_keep[1] = f->f_keep[1]; a.f_slot = &f->sv.st->f_slot; }
  if (gate) { a.cap = &f->sv.st->cap; a.over_flag = f->d_over; }   // (round 5: the capped start applies to the fp64 and the subsample passes too)
  if (ev) MLN_HIP(ctx, hipEventRecord(ev[0], ctx->stream));
  if (f->L32 && (gate || use32)) {
    ObjArgs a32 = a;
    a32.L32 = f->L32;
    a32.l32_fixed = f->l32_fixed;
    a32.gate_want = MLN_GATE_F32;
    if (gate) a32.cap = &f->sv.st->cap;
    MLN_TRY(launch_objective(ctx, a32));
  }
  if (ev) MLN_HIP(ctx, hipEventRecord(ev[1], ctx->stream));
  if (gate || !use32 || !f->L32) {
    a.gate_want = MLN_GATE_F64;
    MLN_TRY(launch_objective(ctx, a));
  }
  if (gate && sub_strides) {
    // the subsample objectives of the solver's first phase: the same fp64 kernel over every s-th row, one launch per
    // level (the solver's state says which one works; same grid: workgroups past the shorter row range write zero
    // partials), partial sums scaled by s
    for (size_t lv = 0; lv < sub_strides->size(); ++lv) {
      const int64_t sub_stride = (*sub_strides)[lv];
      ObjArgs as = a;
      int64_t first = 0, rows = 0;
      fit_sample_rows(f, sub_stride, &first, &rows);
      as.n = rows; as.row_first = first; as.row_stride = sub_stride; as.out_scale = (double)sub_stride;
      as.f_keep[0] = as.f_keep[1] = nullptr; as.f_slot = nullptr;
      as.gate_want = MLN_GATE_SUB;
      as.gate2 = &f->sv.st->sub_level; as.gate2_want = (int)lv;
      MLN_TRY(launch_objective(ctx, as));
    }
  }
  if (ev) MLN_HIP(ctx, hipEventRecord(ev[2], ctx->stream));
  MLN_TRY(launch_reduce_obj2(ctx, a, f->d_zr + ld2 + m, f->d_zr + ld2));
  MLN_TRY(dev_allreduce(ctx, f->d_zr + ld2, m + 2));
  if (f->kspace) {
    GemvTri g2{f->Cinv, ld, m, f->d_zr, gn_dev, nullptr, 0, m, m, 0, 0, gate};      // g_u = R^-1 (q + r)
    g2.xadd = f->d_zr + ld2;
    MLN_TRY(launch_gemv_tri(ctx, g2));
  } else {
    GemvTri g2{f->Q2, 2 * ld, m, f->d_zr, gn_dev, nullptr, 0, m, m, ld, ld2, gate};
    MLN_TRY(launch_gemv_tri(ctx, g2));                                 // [C^-1 | C^-1] [z ; r]: lower triangular blocks
  }
  return MLN_OK;
}

// host-synchronous form (SciPy-driven route, mln_objective_precond)
int fit_objective_u(mln_fit* f, const double* u, double* loss, double* grad_u, double* z_out,
                           bool use32) {
  mln_ctx* ctx = f->ctx;
  const int64_t m = f->m;
  MLN_HIP(ctx, hipMemcpyAsync(f->d_u, u, sizeof(double) * m, hipMemcpyDefault, ctx->stream));
  hipEvent_t ev[3] = {f->ev0, use32 ? f->ev1 : f->ev0, f->ev1};
  MLN_TRY(fit_enqueue_eval(f, f->d_u, f->d_gu, use32, nullptr, ev));
  MLN_HIP(ctx, hipMemcpyAsync(f->h_out, f->d_zr + f->ld2 + m, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MLN_HIP(ctx, hipMemcpyAsync(f->h_out + 1, f->d_gu, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
  // prior: 1/2 |z|^2 (explicit) = 1/2 w . (Kj w) (implicit: d_zr holds q = Kj w, d_w holds w)
  MLN_HIP(ctx, hipMemcpyAsync(f->h_z, f->kspace ? f->d_w : f->d_zr, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
  if (f->kspace) MLN_HIP(ctx, hipMemcpyAsync(f->h_out + 1 + m, f->d_zr, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
  MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  obj_account(f, use32);
  double zz = 0.0;
  for (int64_t j = 0; j < m; ++j) zz += f->h_z[j] * (f->kspace ? f->h_out[1 + m + j] : f->h_z[j]);
  *loss = f->h_out[0] + 0.5 * zz + 0.5 * (double)m * std::log(2.0 * M_PI);
  std::memcpy(grad_u, f->h_out + 1, sizeof(double) * m);
  if (z_out && f->kspace) {                       // z = C^-T u = Lp^T w
    MLN_TRY(fit_ensure_lp(f, false));
    MLN_TRY(fit_small_gemv(f, f->Lp, 1, f->d_w, f->d_z));
    MLN_HIP(ctx, hipMemcpyAsync(z_out, f->d_z, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  } else if (z_out) {
    std::memcpy(z_out, f->h_z, sizeof(double) * m);
  }
  return MLN_OK;
}

extern "C" int mln_objective_precond(mln_fit* f, const double* u, double* loss, double* grad_u, double* z_out) {
  if (!f || !u || !loss || !grad_u) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  if (!f->V) { mln_set_error(ctx, "mln_fit_set_likelihood has not been called"); return MLN_ERR_ARG; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  MLN_TRY(fit_build_precond(f, 1));
  return fit_objective_u(f, u, loss, grad_u, z_out);
}

// ---- a-8: the MAP solve: device-resident L-BFGS (solver.hip) --------------------------------------------------
// Reference: inference.minimize_lbfgsb (inference.py:272-288) = SciPy L-BFGS-B without bounds.  Same method
// (limited-memory BFGS two-loop recursion, H0 = s.y / y.y, sufficient-decrease backtracking from step 1 -- Armijo
// only: SciPy's dcsrch also enforces the curvature condition, so iteration counts are not comparable one to one) and
// SciPy's stopping tests (relative decrease <= ftol, max|g| <= gtol, maxiter) on the preconditioned variable u.
// The optimiser's vectors and decisions live on the device; the host only enqueues evaluation chains in batches
// and looks at the solver's state once per batch.
int fit_solver_alloc(mln_fit* f, int maxcor) {
  mln_ctx* ctx = f->ctx;
  if (f->sv_block && f->sv_maxcor >= maxcor) return MLN_OK;
  if (f->sv_block) { MLN_HIP(ctx, hipStreamSynchronize(ctx->stream)); MLN_HIP(ctx, mln_dfree(f->sv_block)); f->sv_block = nullptr; }
  const size_t ld = (size_t)f->ldl;
  const size_t n_dbl = 6 * ld + 2 * (size_t)maxcor * ld + 2 * 64 + 4 * 512 + (sizeof(SolverState) + 63) / 64 * 8;
  MLN_HIP(ctx, mln_dmalloc(&f->sv_block, sizeof(double) * n_dbl));
  MLN_HIP(ctx, hipMemsetAsync(f->sv_block, 0, sizeof(double) * n_dbl, ctx->stream));
  double* p = (double*)f->sv_block;
  SolverBuffers& b = f->sv;
  b.u = p; p += ld; b.g = p; p += ld; b.un = p; p += ld; b.gn = p; p += ld; b.d = p; p += ld;
  b.S = p; p += (size_t)maxcor * ld; b.Y = p; p += (size_t)maxcor * ld;
  b.rho = p; p += 64; b.yy = p; p += 64;
  b.c = p; p += ld;
  b.trace = p; p += 4 * 512;
  b.st = (SolverState*)p;
  b.ld = (int64_t)ld;
  b.z = f->kspace ? f->d_w : f->d_zr;
  b.z2 = f->kspace ? f->d_zr : nullptr;      // implicit mode: prior = 1/2 w . (Kj w)
  b.lik = f->d_zr + f->ld2 + f->m;
  b.over = f->d_zr + f->ld2 + f->m + 1;
  f->sv_maxcor = maxcor;
  if (!f->h_state) MLN_HIP(ctx, mln_hmalloc((void**)&f->h_state, sizeof(SolverState)));
  return MLN_OK;
}

extern "C" int mln_map_solve(mln_fit* f, const double* z0, const mln_solver_opts* opts_in, double* z_out,
                             double* loss_out, int32_t* n_eval_out, int32_t* n_iter_out, int32_t* status_out) {
  if (!f || !z0 || !z_out) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  if (!f->V) { mln_set_error(ctx, "mln_fit_set_likelihood has not been called"); return MLN_ERR_ARG; }
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  if (f->m > 8192) {
    mln_set_error(ctx, "mln_map_solve: the device-resident solver holds at most 8192 landmarks; drive mln_objective_precond "
                       "from the host instead (the Python binding does: inference.minimize_lbfgsb)");
    return MLN_ERR_UNSUPPORTED;
  }
  MLN_TRY(fit_build_precond(f, 1));
  fit_events_borrow(ctx, &f->evs);       // (handed back by fit_free, with the handle)
  mln_solver_opts o = {5000, 10, 30, 1e-13, 1e-7};
  if (opts_in) o = *opts_in;
  if (o.maxcor < 1) o.maxcor = 1;
  if (o.maxcor > 64) o.maxcor = 64;
  if (o.maxls < 1) o.maxls = 1;
  const int64_t m = f->m;
  MLN_TRY(fit_solver_alloc(f, o.maxcor));
  for (int b = 0; b < 2; ++b)
    if (!f->f_keep[b]) MLN_HIP(ctx, mln_dmalloc((void**)&f->f_keep[b], sizeof(double) * (size_t)(f->n > 0 ? f->n : 1)));
  f->f_final = -1;
  // u0 = C^T z0, identical on every rank (implicit mode: C^T z0 = R^T (Lp^-T z0), api_precond.hip fit_factor_precond)
  MLN_HIP(ctx, hipMemcpyAsync(f->d_u, z0, sizeof(double) * m, hipMemcpyDefault, ctx->stream));
  if (f->kspace) {
    MLN_TRY(fit_w_from_z(f, f->d_u, f->d_w, is_device_ptr(z0) ? nullptr : z0));
    MLN_TRY(fit_small_gemv(f, f->C, 1, f->d_w, f->d_gu));
  } else {
    MLN_TRY(fit_small_gemv(f, f->C, 1, f->d_u, f->d_gu));
  }
  MLN_TRY(dev_bcast0(ctx, f->d_gu, m));
  // Mixed precision: while an fp32 copy of the n x m buffer exists, the first passes stream it (half the bytes);
  // the solver switches to the fp64 buffer by itself (see k_solver_step) and finishes at the same tolerances as
  // a pure fp64 run.
  const int trace_lvl = std::getenv("MELLON_AMD_TRACE") ? std::atoi(std::getenv("MELLON_AMD_TRACE")) : 0;
  const bool phase32 = f->kspace && f->L32 != nullptr;
  SolverState init{};
  init.gate = phase32 ? MLN_GATE_F32 : MLN_GATE_F64;
  init.mode = MLN_SOLVE_FIRST;
  init.status = 1;
  init.maxiter = o.maxiter; init.maxcor = o.maxcor; init.maxls = o.maxls;
  init.m = (int)m;
  init.ftol = o.ftol; init.gtol = o.gtol;
  // progress per iteration below which the 32-bit surrogate is left for the fp64 buffer (relative to the loss):
  // the fp32 copy's optimum sits ~1e-5 (relative loss) from the true one, the fixed-point copy's ~1e-9
  init.ftol32 = f->l32_fixed ? 1e-9 : 3e-6;
  // ... and after that first fp64 evaluation the solve continues on the 32-bit copy WITH its first-order correction
  // (solver.hip), the fp64 objective verifying the final point (MELLON_AMD_CORRECTED=0: finish on the fp64 buffer)
  init.use_corr = (phase32 && f->l32_fixed) ? 1 : 0;
  if (const char* ev = mln_experiment("MELLON_AMD_CORRECTED")) init.use_corr = init.use_corr && std::atoi(ev) != 0;
  init.prior_const = 0.5 * (double)m * std::log(2.0 * M_PI);
  init.t0 = 1.0;
  init.boost = 0.15;    // solver.hip "step-length memory"; MELLON_AMD_LS_BOOST=0 keeps every first trial at 1
  if (const char* ev = mln_experiment("MELLON_AMD_LS_BOOST")) init.boost = std::atof(ev);
  // capped start (solver.hip): on the 32-bit copy the likelihood's e^t is continued linearly beyond t = 7 while the loss
  // still falls steeply; MELLON_AMD_EXP_CAP=<t> moves the cap, MELLON_AMD_EXP_CAP=off removes it
  init.cap = 9.0;      // (tools/r05_ab_c3.sh, C3 seeds 3-6, mean step: off 149 ms | 5: 148 | 6: 164 | 7: 139 | 8: 140 | 9: 130 | 10: 132 | 12: ~158;
                       //  tree / heavy tails at 1e6 cells, passes: 7: 130 / 86 | 8: 158 / 81 | 9: 158 / 97 -- against 511 / 473 without it)
  init.cap_step = 4.0;
  if (const char* ev = mln_experiment("MELLON_AMD_EXP_CAP"))
    init.cap = (std::strcmp(ev, "off") == 0 || std::atof(ev) <= 0.0) ? __builtin_inf() : std::atof(ev);
  // (the opt-in mixed solve runs WITHOUT the cap: its surrogate / anchor state machine was tuned with the round-2 linear cap,
  //  and with the staged quadratic one the tree of tests/test_gpu_round4.py ran into the iteration limit)
  if (phase32 && !mln_experiment("MELLON_AMD_EXP_CAP")) init.cap = __builtin_inf();
  init.cap0 = init.cap;
  init.boost_fall = 0.15;
  // Subsample start (solver.h): when the preconditioner's Gram came from every s-th cell (s >= 4), the solve starts
  // on the MAP problem of exactly those cells -- the Ridge matrix is ITS Hessian at a = 1 -- at 1/s of the bytes per
  // pass, and moves to all cells once that problem's progress per iteration is below sub_tol.  The walk down from the
  // Ridge start (a dozen passes) then costs about two.  MELLON_AMD_SUBSAMPLE=0 disables.
  // Which cells: ~32 m of them (every (3 s / 16)-th cell for a Gram stride s = n / 6 m; nested levels did not pay).  tools/solver_sweep.py, five data seeds at C3, mean step in ms:
  // no subsample 302 | stride 16: 241 | 12: 204 | 8: 203 | 6: 193 | 4: 203 | 16 then 8: 213 | 16 then 4: 215.
  // The smaller the sample, the cheaper its passes but the more its optimum overfits (at stride 16 the first full
  // evaluation finds the loss 60 % above the optimum's and e^{f+V} of unseen cells up to 1e5).
  std::vector<int64_t> sub_strides;
  if (f->precond_stride >= 11) sub_strides.push_back(std::max<int64_t>(2, 3 * f->precond_stride / 16));
  if (const char* ev = mln_experiment("MELLON_AMD_SUBSAMPLE")) { if (std::atoi(ev) == 0) sub_strides.clear(); }
  const std::vector<int64_t>* subs = sub_strides.empty() ? nullptr : &sub_strides;
  init.gate_full = init.gate;
  init.sub_tol = 1e-3;
  init.sub_max_evals = 1 << 30;  // (off: the tree of tools/hard_cases.py spends 139 evaluations there -- and needs MORE passes in total with a limit of 48 or 32)
  init.n_sub_levels = (int)sub_strides.size();
  init.sub_level = 0;
  if (subs) init.gate = MLN_GATE_SUB;
  // Preconditioner rebuild (solver.h, precond_rebuild.hip): pays when the evaluations it saves (measured: 33-40 full
  // passes without it, 15-26 with it) cost more than the m^3 work of a second factorisation -- decided from rank 0's
  // measurement of the first build, the same on every rank.  An evaluation = one pass of this rank's rows + ~0.14 ms of
  // small launches.  MELLON_AMD_REBUILD=0 / 1 forces the decision.
  const double pass_s = (double)f->n * (double)f->ldl * 8.0 / 6.5e12 + 1.4e-4;
  // (Emulated ranks of C3, tools/emulate_rank.py.  Until round 4 the emulation drew the WHOLE importance sample from rank 0's
  //  shard -- 30 000 rows at 8 ranks where a real rank contributes 3 750 -- and so priced the rebuild at 15 ms there, a loss of
  //  2.5 ms per step; with the rank's share it costs 10.9 ms and the step falls from 55.2 to 50.7 ms.  The solve without it
  //  needs 29-33 full passes at C3, with it 14: the rebuild is worth ~15 passes, and the threshold is 13 of them -- on at
  //  8 ranks (11.8 ms of passes against 10.9 ms of build), off from 16 ranks on (6.8 against 10.4).)
  // The price of a build comes from a MODEL of it, not from the stopwatch on the first one: a measured time made the
  // decision -- and with it the iteration path and the last digits of the result -- depend on whether the process was warm
  // (the first fit of a process: no rebuild, 11 evaluations; every later one: rebuild, 8 evaluations, log-density 2e-5 off
  // at the default stopping rule on a 100-cell problem).  Model: two factorisation-like chains of m / 128 dependent block
  // steps (0.25 ms per block: chol(C'), inverses, operators), the integer Gram of this rank's sample rows at 1 POP/s, the
  // whitening's 2 m^3 flops (column-split over >= 3 ranks) at 45 TFLOP/s -- 18.5 / 12.2 / 11.1 ms at C3 on 1 / 4 / 8 ranks
  // against the measured 18.7 / 13.6 / 10.9.
  // Round 5: the Gram is no longer whitened (w-space factor, api_precond.hip): a build is the Gram of this rank's sample rows
  // plus ONE factorisation-like chain with its inverse and the triangular product C^-1 = R^-1 Lp.
  const double md = (double)f->m;
  const double gram_rows = (double)f->n / (double)(f->precond_stride > 0 ? f->precond_stride : 1);
  const double build_model_s = 2.2e-4 * std::ceil(md / 128.0) + gram_rows * md * md * 2.0 / 1.0e15;
  double want_rebuild = (f->build_seconds > 0.0 && 13.0 * pass_s > build_model_s) ? 1.0 : 0.0;
  if (const char* ev = mln_experiment("MELLON_AMD_REBUILD")) want_rebuild = std::atoi(ev) != 0 ? 1.0 : 0.0;
  if (phase32 && !(f->l32_fixed)) want_rebuild = 0.0;   // (mixed solves pause at their fp64 anchor, which only the corrected fixed-point surrogate has)
  // The rebuild reads the rows' f of the last accepted pass (f_keep), which a rank only has while its shard fits the
  // kernel's f staging: with uneven or very large shards that is a per-rank fact, and the branch at the pause issues
  // collectives (Gram all-reduce, the sample's global sum) -- so the decision is made ONCE, here, for all ranks: rank 0's
  // cost rule AND every rank able to keep f (one all-reduce of two numbers: rank 0's vote, the count of ranks that cannot).
  const bool keeps_f = f->f_keep[0] && f->f_keep[1] && objective_can_keep_f(f->n, f->n_wg);
  {
    double vote[2] = {ctx->rank == 0 ? want_rebuild : 0.0, keeps_f ? 0.0 : 1.0};
    MLN_HIP(ctx, hipMemcpyAsync(f->d_tmp, vote, sizeof(vote), hipMemcpyHostToDevice, ctx->stream));
    MLN_TRY(dev_allreduce(ctx, f->d_tmp, 2));
    MLN_HIP(ctx, hipMemcpyAsync(vote, f->d_tmp, sizeof(vote), hipMemcpyDeviceToHost, ctx->stream));
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    want_rebuild = (vote[0] != 0.0 && vote[1] == 0.0) ? 1.0 : 0.0;
  }
  // (round 5: up to max_rebuilds of them -- after the first, whenever the solve has fallen back to slow linear convergence,
  //  solver.hip; C3 takes one)
  int max_rebuilds = 1000;    // (rate-limited by the solver's own rules: six iterations apart, only on slow linear convergence)
  if (const char* ev = mln_experiment("MELLON_AMD_MAX_REBUILDS")) max_rebuilds = std::max(1, std::atoi(ev));
  init.rebuild_armed = want_rebuild != 0.0 ? (phase32 ? 1 : max_rebuilds) : 0;      // (the opt-in mixed solve keeps its single rebuild)
  init.it_resume = -1;
  // an overshooting start (solver.h over_many): more than one cell in 10 000 above the cap
  init.over_many = std::max(8.0, 1e-4 * (double)f->n * (double)(ctx->n_ranks > 1 ? ctx->n_ranks : 1));
  init.over_cnt_acc = 0.0;
  // (a rebuild right at the switch from the subsample to all cells was measured in round 3: 37-96 full passes -- the unseen
  //  cells' weights are still too wild there; the knob is gone)
  init.switch_t0 = 0.35;
  init.gap_tol = 0.2 * o.ftol;     // (tools/solver_sweep.py, seven data seeds at C3: 15.9 -> 14.7 full passes with both rules, log-density
                                   //  within 4e-8 of the old stop -- the spread between two runs of the old rule; 0.5 ftol: 14.3 passes, 1.8e-7)
  init.dec_prev = 0.0; init.dec_prev2 = 0.0;
  init.rebuild_tol = 1e-3;       // (tools/solver_sweep.py at C3, two seeds: 1e-2 -> 23-28 full passes, 1e-3 -> 20-22, 2e-4 -> 22-26)
  init.start_cap = 1e30;         // (solver.h: pathological start; C3's Ridge start evaluates to 4.8e9)
  init.n_shrink = 0;
  init.revert_after = 0; init.it_at_resume = -1; init.pause_reason = 0;
  int revert_after = 60;         // accepted iterations a rebuilt preconditioner gets to finish the solve (C3: ~9)
  if (const char* ev = mln_experiment("MELLON_AMD_REVERT_AFTER")) revert_after = std::atoi(ev);
  double rebuild_rows_per_m = 6.0;   // (first rebuild -- C3: 6 m, 12 m, 24 m importance-sampled rows give the same pass counts, 6 m is the cheapest Gram;
                                     //  the later ones, which only slow solves reach, take twice as many: tree 169 -> 150, 115 -> 79, heavy tails 128 -> 104 passes)
  MLN_TRY(launch_solver_init(ctx, f->sv, init, f->d_gu));
  const int* gate = &f->sv.st->gate;
  static const bool timing = !(std::getenv("MELLON_AMD_TIMING") && std::atoi(std::getenv("MELLON_AMD_TIMING")) == 0);
  int n_enq = 0, rebuilds_this_solve = 0, failed_rebuilds = 0;
  auto events_for = [&](int i) -> hipEvent_t* {
    if (!timing || i >= 512) return nullptr;
    while ((int)f->evs.size() < 3 * (i + 1)) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      f->evs.push_back(e);
    }
    return &f->evs[3 * i];
  };
  MLN_TRY(fit_enqueue_eval(f, f->sv.un, f->sv.gn, false, gate, events_for(n_enq), subs));
  ++n_enq;
  // evaluation t of the solver's trace ran in enqueue slot t + shift: the chains left in a batch after a pause are
  // no-ops that use up slots (their events time nothing)
  std::vector<std::pair<int, int>> slot_shift;      // (first trace index, shift)
  const std::vector<int64_t>* subs_live = subs;
  int batch = 8;
  const int64_t hard_cap = (int64_t)o.maxiter * o.maxls + 16;
  for (;;) {
    for (int b = 0; b < batch; ++b) {
      MLN_TRY(launch_solver_step(ctx, f->sv, (int)m));
      MLN_TRY(fit_enqueue_eval(f, f->sv.un, f->sv.gn, false, gate, events_for(n_enq), subs_live));
      ++n_enq;
    }
    // rank 0's state decides for everyone (it is the same state on every rank by construction: identical inputs,
    // identical all-reduced sums, deterministic kernels -- this only rules out a hang should that ever fail)
    MLN_TRY(dev_bcast0(ctx, (double*)f->sv.st, (int64_t)(sizeof(SolverState) / sizeof(double))));
    MLN_HIP(ctx, hipMemcpyAsync(f->h_state, f->sv.st, sizeof(SolverState), hipMemcpyDeviceToHost, ctx->stream));
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (f->h_state->gate == MLN_GATE_DONE) break;
    // the subsample phase never comes back: its (gated-off, ~4 us each) launches need not ride along any more
    if (subs_live && f->h_state->n_eval_sub > 0 && f->h_state->gate != MLN_GATE_SUB) subs_live = nullptr;
    if (f->h_state->gate == MLN_GATE_PAUSE) {
      // ---- second preconditioner at the accepted point (whose rows' f the last accepted fp64 pass left in f_keep) ----
      const double tr0 = now_s(), ex_r0 = f->emu_excluded;
      const SolverState ps = *f->h_state;
      const bool revert = ps.pause_reason == 2 && f->saved_precond[0] != nullptr;
      if (!revert && (ps.pause_reason == 2 || !ps.f_valid)) {
        // (f_valid is a function of the solver's state, identical on every rank; keeping f was settled collectively above)
        // no per-row f to weight the cells with / nothing to go back to: resume with the preconditioner we have
        MLN_TRY(launch_solver_resume(ctx, f->sv, ps.gate_after_pause, 0));
      } else {
        double *zt = nullptr, *gz = nullptr, *cz = nullptr;
        MLN_HIP(ctx, mln_dmalloc((void**)&zt, sizeof(double) * 3 * (size_t)f->ldl));
        gz = zt + f->ldl; cz = gz + f->ldl;
        MLN_HIP(ctx, hipMemsetAsync(zt, 0, sizeof(double) * 3 * (size_t)f->ldl, ctx->stream));
        // old variable -> z-space:  z = C^-T u,  g_z = C g_u  (and the surrogate's correction c, a gradient in u, likewise);
        // implicit mode -> w-space (w = Lp^-T z):  w = P u = R^-T u,  g_w = R g_u   (f->C holds R there, P^T = R^-1)
        int rc = f->kspace ? fit_small_gemv(f, f->P, 0, f->sv.u, zt) : fit_small_gemv(f, f->Cinv, 1, f->sv.u, zt);
        if (rc == MLN_OK) rc = fit_small_gemv(f, f->C, 0, f->sv.g, gz);
        if (rc == MLN_OK && ps.corr) rc = fit_small_gemv(f, f->C, 0, f->sv.c, cz);
        // (The curvature pairs could be carried through the change of variable -- s' = T s, y' = T^-T y with T = C'^T C^-T --
        //  but that measured 1-3 full passes WORSE than a fresh history at C3, five data seeds: the new factor already holds
        //  the curvature the old pairs describe, relative to a metric that is gone.  They are dropped.)
        int outcome = 0;
        if (rc == MLN_OK) rc = revert ? fit_precond_revert(f) : fit_rebuild_precond(f, f->f_keep[ps.f_slot], rebuild_rows_per_m * (rebuilds_this_solve > 0 ? 2.0 : 1.0), &outcome, ps.cap);
        if (rc == MLN_OK && outcome == 0) {
          // z-space -> new variable:  u = C^T z,  g_u = C^-1 g_z      (w-space:  u = R^T w,  g_u = R^-1 g_w = P^T g_w)
          rc = fit_small_gemv(f, f->C, 1, zt, f->sv.u);
          if (rc == MLN_OK) rc = f->kspace ? fit_small_gemv(f, f->P, 1, gz, f->sv.g) : fit_small_gemv(f, f->Cinv, 0, gz, f->sv.g);
          if (rc == MLN_OK && ps.corr) rc = f->kspace ? fit_small_gemv(f, f->P, 1, cz, f->sv.c) : fit_small_gemv(f, f->Cinv, 0, cz, f->sv.c);
        }
        (void)hipStreamSynchronize(ctx->stream);
        (void)mln_dfree(zt);
        MLN_TRY(rc);
        if (outcome == 0) {
          // a rebuilt preconditioner is on trial (solver.h: revert_after); a restored one is not.  A mixed solve whose
          // rebuilt preconditioner failed its trial also forgets its anchor (taken at that rebuild's pause point: clip_sweep
          // on the tree ran the corrected surrogate to the iteration limit) and goes on with the plain 32-bit surrogate.
          if (revert && ps.corr && (ps.gate_after_pause & 3) == MLN_GATE_F32)
            MLN_TRY(launch_solver_resume_plain32(ctx, f->sv, MLN_GATE_F32, (int)m, 1));
          else
            // (the trial applies to the LAST rebuild the solve is allowed: while more remain, a stall leads to the next one)
            MLN_TRY(launch_solver_resume(ctx, f->sv, ps.gate_after_pause, 1, (revert || ps.rebuild_armed > 0) ? 0 : revert_after));
          if (revert) f->n_revert += 1; else { f->n_rebuild += 1; rebuilds_this_solve += 1; }
        } else {
          // the rebuild declined (weights too wild) or lost positive definiteness: same variable, same history, carry on --
          // in the mixed solve WITHOUT the anchor just taken: the pause came far from the optimum (that is what the weights
          // say), where the fp64 objective and its 32-bit surrogate differ by more than a first-order correction mends (heavy
          // tails at 1e6 cells: 1e15 in the loss; the corrected surrogate then ran to the iteration limit).  The solve
          // continues on the plain surrogate and anchors when THAT has converged, as a mixed solve without rebuild does.
          // (two attempts that came to nothing: the solve goes on without asking again)
          failed_rebuilds += 1;
          if (ps.corr && (ps.gate_after_pause & 3) == MLN_GATE_F32) MLN_TRY(launch_solver_resume_plain32(ctx, f->sv, MLN_GATE_F32, (int)m));
          else MLN_TRY(launch_solver_resume(ctx, f->sv, ps.gate_after_pause, 0, 0, failed_rebuilds >= 2 ? 0 : -1));
          f->n_rebuild_skipped += 1;
        }
        if (trace_lvl) fprintf(stderr, "[trace] map_solve pause at evaluation %d: %s\n", ps.n_eval,
                               revert ? "second preconditioner failed its trial: first one restored"
                                      : (outcome == 0 ? "preconditioner rebuilt" : (outcome == 1 ? "rebuild declined (weight range)" : "rebuild lost positive definiteness: kept the first")));
      }
      MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
      f->times_rebuild += now_s() - tr0 - (f->emu_excluded - ex_r0);
      slot_shift.push_back({ps.n_eval, n_enq - ps.n_eval});
      continue;
    }
    if (n_enq > hard_cap) { mln_set_error(ctx, "map_solve: the device solver did not terminate"); return MLN_ERR_NOCONV; }
    if (batch < 16 && f->h_state->gate == MLN_GATE_F64) batch = std::min(batch, 6);
  }
  const SolverState st = *f->h_state;
  fit_precond_saved_free(f);
  f->n_start_halvings = st.n_shrink;
  // kernel-time accounting from the per-evaluation events (the pass the solver did not select is a ~2 us no-op)
  std::vector<double> tr;
  const int n_done = st.n_eval < 512 ? st.n_eval : 512;
  if (n_done > 0) {
    tr.resize((size_t)4 * n_done);
    MLN_HIP(ctx, hipMemcpy(tr.data(), f->sv.trace, sizeof(double) * 4 * n_done, hipMemcpyDeviceToHost));
    for (int i = 0; i < n_done && timing; ++i) {
      const int gcode = (int)tr[4 * i + 3] & 15, lvl = (int)tr[4 * i + 3] >> 4;
      const bool was32 = (gcode & 3) == MLN_GATE_F32, was_sub = gcode == MLN_GATE_SUB;
      int slot = i;
      for (const auto& sh : slot_shift) if (i >= sh.first) slot = i + sh.second;
      if (3 * (slot + 1) > (int)f->evs.size()) continue;
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, f->evs[3 * slot + (was32 ? 0 : 1)], f->evs[3 * slot + (was32 ? 1 : 2)]) != hipSuccess) continue;
      if (was_sub) {
        f->times_sub += 1e-3 * ms; f->evals_sub += 1;
        f->sub_pass_equiv += 1.0 / (double)((lvl >= 0 && lvl < (int)sub_strides.size()) ? sub_strides[lvl] : 1);
      }
      else if (was32) { f->times32 += 1e-3 * ms; f->evals32 += 1; }
      else { f->times[5] += 1e-3 * ms; f->times[6] += 1.0; f->times[7] = (double)f->n * (double)f->ldl * 8.0; }
    }
    if (trace_lvl >= 2)
      for (int i = 0; i < n_done; ++i)
        fprintf(stderr, "[eval %d] %s mode=%d t=%.3g f=%.15g\n", i, ((int)tr[4 * i + 3] & 15) == MLN_GATE_F32 ? "f32" : (((int)tr[4 * i + 3] & 15) == MLN_GATE_F32C ? "f32c" : (((int)tr[4 * i + 3] & 15) == MLN_GATE_SUB ? (((int)tr[4 * i + 3] >> 4) ? "sub1" : "sub0") : "f64")),
                (int)tr[4 * i + 2], tr[4 * i + 1], tr[4 * i]);
  }
  // z = C^-T u and w = P u at the accepted point, remembered for transform / predictor weights
  {
    if (f->kspace) {                              // w = R^-T u, z = Lp^T w
      GemvTri g1{f->P, f->ldl, m, f->sv.u, f->d_w_cached, nullptr, 1, m, m, 0, 0, nullptr};
      MLN_TRY(launch_gemv_tri(ctx, g1));
      MLN_TRY(fit_ensure_lp(f, false));
      MLN_TRY(fit_small_gemv(f, f->Lp, 1, f->d_w_cached, f->d_z));
    } else {
      GemvTri g1{f->Q1, f->ldl, m, f->sv.u, f->d_z, nullptr, 1, m, m, 0, 0, nullptr};
      MLN_TRY(launch_gemv_tri(ctx, g1));
    }
    f->z_cached.assign((size_t)m, 0.0);
    MLN_HIP(ctx, hipMemcpyAsync(f->z_cached.data(), f->d_z, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
    MLN_HIP(ctx, hipMemcpyAsync(z_out, f->d_z, sizeof(double) * m, hipMemcpyDefault, ctx->stream));
    MLN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (st.f_valid && objective_can_keep_f(f->n, f->n_wg)) f->f_final = st.f_slot;   // f = L z + mu at this z is already there (mln_transform)
  if (trace_lvl)
    fprintf(stderr, "[trace] map_solve: %d evaluations (%d on the 32-bit copy, %d on the row subsample of stride %lld), %d iterations, "
            "%d rebuild(s), %d enqueued, status %d\n", st.n_eval, st.n_eval32, st.n_eval_sub, (long long)(subs ? sub_strides[0] : 0), st.it,
            f->n_rebuild, n_enq, st.status);
  if (loss_out) *loss_out = st.fx;
  if (n_eval_out) *n_eval_out = st.n_eval;
  if (n_iter_out) *n_iter_out = st.it;
  // A solve that stops on its iteration limit / an exhausted line search while the accepted point still has rows above the
  // likelihood cap reports the loss and gradient of the capped minorant, not of inference.py's objective: bit 2 of the
  // status says so (a converged solve never ends capped: k_solver_step raises the cap and re-evaluates first).
  if (status_out) *status_out = st.status | ((st.status != 0 && st.over_acc && st.cap < 1e300) ? 4 : 0);
  return MLN_OK;
}

extern "C" int mln_weights_cholesky(mln_fit* f, const double* z, double* w) {
  if (!f || !z || !w) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  if (!f->Lp) { mln_set_error(ctx, "this fit handle holds no Lp"); return MLN_ERR_ARG; }
  DevOut o;
  MLN_TRY(o.init(ctx, w, (size_t)f->m));
  if (f->kspace && !is_device_ptr(z) && f->z_cached.size() == (size_t)f->m &&
      std::memcmp(z, f->z_cached.data(), sizeof(double) * f->m) == 0) {
    MLN_HIP(ctx, hipMemcpyAsync(o.dev, f->d_w_cached, sizeof(double) * f->m, hipMemcpyDeviceToDevice, ctx->stream));
    return o.commit();
  }
  MLN_TRY(fit_ensure_lp(f));
  MLN_HIP(ctx, hipMemcpyAsync(o.dev, z, sizeof(double) * f->m, hipMemcpyDefault, ctx->stream));
  MLN_TRY(triinv_solve_left_T(ctx, f->tri, o.dev, 1, 1));  // conditional.py:818
  return o.commit();
}

extern "C" int mln_weights_full(mln_fit* f, const double* y, int64_t p, double mu, double* w) {
  if (!f || !y || !w || p < 1) return MLN_ERR_ARG;
  mln_ctx* ctx = f->ctx;
  MLN_HIP(ctx, hipSetDevice(ctx->device));
  if (!f->Lp) { mln_set_error(ctx, "this fit handle holds no Lp"); return MLN_ERR_ARG; }
  MLN_TRY(fit_ensure_lp(f));
  const int64_t cnt = f->m * p;
  DevOut o;
  MLN_TRY(o.init(ctx, w, (size_t)cnt));
  MLN_HIP(ctx, hipMemcpyAsync(o.dev, y, sizeof(double) * cnt, hipMemcpyDefault, ctx->stream));
  // r = y - mu ; w = Lp^-T Lp^-1 r                                conditional.py:263-264
  if (mu != 0.0) {
    double* ones = nullptr;
    MLN_HIP(ctx, mln_dmalloc((void**)&ones, sizeof(double) * cnt));
    std::vector<double> h((size_t)cnt, 1.0);
    MLN_HIP(ctx, hipMemcpyAsync(ones, h.data(), sizeof(double) * cnt, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_axpby(ctx, cnt, -mu, ones, 1.0, o.dev);
    (void)hipStreamSynchronize(ctx->stream);
    (void)mln_dfree(ones);
    if (rc != MLN_OK) return rc;
  }
  MLN_TRY(triinv_solve_left(ctx, f->tri, o.dev, p, p));
  MLN_TRY(triinv_solve_left_T(ctx, f->tri, o.dev, p, p));
  return o.commit();
}

extern "C" int mln_stage_times(mln_fit* f, double* out) {
  if (!f || !out) return MLN_ERR_ARG;
  for (int i = 0; i < 8; ++i) out[i] = f->times[i];
  out[8] = f->times32;                                  // 32-bit warm-up passes: kernel seconds (HIP events)
  out[9] = (double)f->evals32;                          //                        launches
  out[10] = f->L32 ? (f->l32_fixed ? 2.0 : 1.0) : 0.0;  //                        format of the copy
  out[11] = f->emu_excluded;                            // MELLON_AMD_EMULATE_RANKS: seconds spent on other ranks' blocks
  out[12] = f->times_sub;                               // subsample passes of the solver's first phase: kernel seconds
  out[13] = (double)f->evals_sub;                       //                                                launches
  out[14] = (double)(f->precond_stride > 0 ? f->precond_stride : 1);   // their row stride (= the Gram sample's)
  out[15] = f->times_rebuild;                           // second preconditioner: wall seconds (selection, Gram, factorisation)
  out[16] = (double)f->n_rebuild;
  // passes over the n x m buffer in full-fp64-pass equivalents (bytes streamed / bytes of one fp64 pass)
  out[17] = f->times[6] + 0.5 * (double)f->evals32 + f->sub_pass_equiv;
  out[18] = (double)f->n_rebuild_skipped;               // rebuilds declined (weight range) or failed (not positive definite)
  out[19] = (double)f->n_revert;                        // second preconditioner failed its trial: first one restored
  out[20] = (double)f->n_start_halvings;                // halvings of a start whose loss was not finite or above 1e30
  out[21] = (double)f->rank_path;                       // mln_fit_gram_rank: 1 = inertia, 2 = tridiagonalisation
  return MLN_OK;
}

