// The MAP solve's optimiser as ONE single-workgroup kernel per evaluation (reference: inference.minimize_lbfgsb,
// inference.py:272-288: SciPy L-BFGS-B without bounds behind jaxopt.ScipyMinimize).
//
// Same method as before -- limited-memory BFGS two-loop recursion with H0 = s.y / y.y, sufficient-decrease
// backtracking from step 1, SciPy's stopping tests (relative decrease <= ftol, max|g| <= gtol, maxiter) -- on the
// preconditioned variable u (z = C^-T u).  What changed is where it runs: the m-vectors (u, g, the <= 30 curvature
// pairs) and every decision (accept / backtrack / switch from the fp32 copy to the fp64 buffer / stop) stay on the
// device.  One evaluation is then a fixed chain of launches
//     k_solver_step -> [z ; w] = Q1 un -> k_objective32 | k_objective (gated) -> reduction -> (all-reduce) -> gn = Q2 [z ; r]
// which the host enqueues in batches without ever waiting for a result; after the solver has set its gate to DONE
// the remaining launches of a batch return immediately.  With cells sharded over ranks every rank runs this same
// optimiser on the same all-reduced numbers, so no rank-0 round trip is needed either.
//
// 512 threads own m <= 8192 elements (EPT each).  Dot products: wave64 shuffle + 8 partials through LDS, summed
// in fixed order by every thread -> the result is uniform across the workgroup and bit-reproducible.
#include <cmath>

#include "objective.h"
#include "solver.h"

namespace {

constexpr int ST = 512;    // threads of the step kernel

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < ST / 64; ++w) s += red[w];
  return s;
}

__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, double (*red3)[ST / 64]) {
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red3[0][threadIdx.x >> 6] = a; red3[1][threadIdx.x >> 6] = b; red3[2][threadIdx.x >> 6] = c; }
  __syncthreads();
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int w = 0; w < ST / 64; ++w) { s0 += red3[0][w]; s1 += red3[1][w]; s2 += red3[2][w]; }
  a = s0; b = s1; c = s2;
}

__device__ __forceinline__ double block_max(double v, double* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < ST / 64; ++w) s = fmax(s, red[w]);
  return s;
}

template <int EPT>
__global__ __launch_bounds__(ST) void k_solver_step(SolverBuffers b) {
  SolverState* st = b.st;
  if ((st->gate & 3) == MLN_GATE_DONE) return;    // DONE or PAUSE
  __shared__ double red[ST / 64];
  __shared__ double red3[3][ST / 64];
  __shared__ double alpha_s[64];
  const int tid = threadIdx.x;
  const int m = st->m;
  const int64_t ld = b.ld;
  bool on[EPT];
  int idx[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) { idx[e] = tid + ST * e; on[e] = idx[e] < m; }
  double u[EPT], g[EPT], un[EPT], gn[EPT], d[EPT];
  double zz = 0.0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    u[e] = on[e] ? b.u[idx[e]] : 0.0;
    g[e] = on[e] ? b.g[idx[e]] : 0.0;
    un[e] = on[e] ? b.un[idx[e]] : 0.0;
    gn[e] = on[e] ? b.gn[idx[e]] : 0.0;
    d[e] = on[e] ? b.d[idx[e]] : 0.0;
    const double zv = on[e] ? b.z[idx[e]] : 0.0;
    const double z2v = b.z2 ? (on[e] ? b.z2[idx[e]] : 0.0) : zv;
    zz = fma(zv, z2v, zz);
  }
  // ---- the evaluation that just finished: loss = 1/2 |z|^2 + (m/2) log 2 pi + likelihood sum (inference.py:45-46,89-91)
  zz = block_sum(zz, red);
  double fn = b.lik[0] + 0.5 * zz + st->prior_const;
  // Which function was that?  phaseA: the plain 32-bit surrogate F32.  phaseC: the corrected one,
  //     F^(u) = F32(u) + c . u + k,   c = grad F(u_a) - grad F32(u_a),   k: F^(u_a) = F(u_a)
  // anchored at the point u_a of the last fp64 evaluation: same values and gradients as the fp64 objective F at u_a, and
  // a Hessian that differs from F's by the 1e-10 relative error of the copy -- its minimiser is F's to second order in
  // |u* - u_a|.  The solver runs phase A until the surrogate's own offset shows (progress <= ftol32), anchors with one
  // fp64 evaluation, converges on F^ by the FINAL tolerances, and then asks the fp64 objective at that point whether
  // F^ told the truth (|F - F^| within the stopping tolerance, or gradient below gtol); if not, that evaluation is
  // the next anchor.  Every fp64 pass saved reads 40 GB less.
  const bool resume = st->mode == MLN_SOLVE_RESUME;   // the host re-expressed u, g in a new preconditioned variable: no evaluation to consume
  const bool phaseA = st->gate == MLN_GATE_F32, phaseC = st->gate == MLN_GATE_F32C, phaseS = st->gate == MLN_GATE_SUB;
  const bool phase32 = phaseA || phaseC;
  const bool approx = phase32 || phaseS;               // the evaluation in flight was of a surrogate (32-bit copy / row subsample)
  if (phaseC) {
    double cu = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const double cv = on[e] ? b.c[idx[e]] : 0.0;
      cu = fma(cv, un[e], cu);
      gn[e] += cv;
    }
    fn += block_sum(cu, red) + st->corr_k;
  }
  int mode = st->mode, it = st->it, n_eval = st->n_eval + (resume ? 0 : 1), n_eval32 = st->n_eval32 + ((phase32 && !resume) ? 1 : 0);
  int n_eval_sub = st->n_eval_sub + ((phaseS && !resume) ? 1 : 0), it_full = st->it_full, sub_level = st->sub_level;
  bool next_level = false;
  int ls = st->ls, k = st->k, head = st->head, status = st->status, gate = st->gate;
  double fx = st->fx, t = st->t, gd = st->gd, corr_k = st->corr_k, t0 = st->t0, cap = st->cap;
  double dec_prev = st->dec_prev, dec_prev2 = st->dec_prev2;
  const bool after_switch = mode == MLN_SOLVE_REEVAL && !phaseS && !phaseC && st->n_eval_sub > 0 && st->it_full == 0;   // (full objective, first direction)
  bool recap = false;
  if (mode != MLN_SOLVE_LS && !resume) t0 = 1.0;
  int f_slot = st->f_slot, f_valid = st->f_valid, corr = st->corr, n_anchor = st->n_anchor;
  int over_acc = st->over_acc;
  double over_cnt_acc = st->over_cnt_acc;
  const double over_cnt_now = (cap < 1e300 && b.over) ? b.over[0] : 0.0;     // rows of the evaluation in flight above the cap
  const int over_now = over_cnt_now > 0.0 ? 1 : 0;
  if (b.trace && tid == 0 && !resume) {
    double* tr = b.trace + 4 * ((n_eval - 1) & 511);
    tr[0] = fn; tr[1] = (mode == MLN_SOLVE_LS) ? t : 0.0; tr[2] = (double)mode; tr[3] = (double)(gate + 16 * st->sub_level);
  }
  bool to_head = false, reeval = false, done = false, verify = false, pause = false, shrink = false;
  int pause_reason = 1;
  if (resume) {
    to_head = true;                      // u, g, fx are the accepted point (in the NEW variable when the history was dropped)
    if (!st->resume_keep_pairs) { k = 0; head = 0; }
    dec_prev = 0.0; dec_prev2 = 0.0;
  } else if (mode == MLN_SOLVE_FIRST && (!isfinite(fn) || fn > st->start_cap) && st->n_shrink < 64) {
    // pathological start (solver.h: start_cap): same gate, same mode, half the point
#pragma unroll
    for (int e = 0; e < EPT; ++e) { u[e] *= 0.5; un[e] = u[e]; }
    shrink = true;
  } else if (mode != MLN_SOLVE_LS) {            // first point, or the same point again on the fp64 buffer
    if (mode == MLN_SOLVE_REEVAL && !phase32 && st->use_corr) {
      // fp64 evaluation at the accepted point u, where fx / g hold the surrogate's loss / gradient (F32 after phase A,
      // F^ after phase C): (re)anchor the correction here -- and, after phase C, let the fp64 gradient decide below
      // whether the solve is over (`verify`)
      verify = corr != 0;
      if (n_anchor < 4) {
        double dcu = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          const double dc = gn[e] - g[e];                 // c_new - c_old  (c_old = 0 after phase A)
          const double cv = (corr && on[e]) ? b.c[idx[e]] : 0.0;
          if (on[e]) b.c[idx[e]] = cv + dc;
          dcu = fma(dc, u[e], dcu);
        }
        dcu = block_sum(dcu, red);
        corr_k = (corr ? corr_k : 0.0) + (fn - fx) - dcu;
        corr = 1; ++n_anchor;
        gate = MLN_GATE_F32C;
      } else {
        corr = 0;                                          // four anchors were not enough: finish on the fp64 buffer
      }
    }
    fx = fn;
    over_acc = over_now; over_cnt_acc = over_cnt_now;
    f_slot ^= 1; f_valid = approx ? 0 : 1;   // the rows' f of this pass belong to the accepted point
#pragma unroll
    for (int e = 0; e < EPT; ++e) g[e] = gn[e];
    // first full fp64 evaluation after the subsample levels: the weights a = e^{f+V} of EVERY cell are on the table and the
    // point is close to the optimum -- the moment for the second preconditioner
    if (mode == MLN_SOLVE_REEVAL && !approx && st->rebuild_armed && corr && n_anchor == 1 && st->use_corr) pause = true;   // the early anchor of the mixed solve
    else to_head = true;
  } else {
    ++ls;
    const bool ok = isfinite(fn) && fn <= fx + 1e-4 * t * gd;
    if (ok) {
      double sy = 0.0, ss = 0.0, yy = 0.0;
      double sv[EPT], yv[EPT];
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        sv[e] = un[e] - u[e];
        yv[e] = gn[e] - g[e];
        sy = fma(sv[e], yv[e], sy); ss = fma(sv[e], sv[e], ss); yy = fma(yv[e], yv[e], yy);
      }
      block_sum3(sy, ss, yy, red3);
      // Step-length memory.  Where some cells' e^{f+V} is still huge (the Ridge start overshoots by up to e^14 in
      // sparse regions) a quasi-Newton step walks down the exponential one unit at a time: the loss falls by ~2x per
      // pass for a dozen passes and the slope along d after the unit step is still ~e^-1 of what it was (a quadratic
      // model promises 0).  So: a first trial that was accepted with more than `boost` of the initial slope left
      // doubles the first trial of the NEXT search (Armijo still decides; a failed trial backtracks as always).
      if (st->boost > 0.0) {
        double sl = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) sl = fma(gn[e], d[e], sl);
        sl = block_sum(sl, red);
        // (only while the loss itself still falls by more than `boost_fall` per pass: later, a doubled trial is a wasted
        //  pass more often than a saved one.  tools/boost_sweep.py, six data seeds at C3: 50.5 -> 41.8 passes on average
        //  with boost = boost_fall = 0.15)
        t0 = (ls == 1 && t >= 1.0 && gd < 0.0 && sl / gd > st->boost && (fx - fn) > st->boost_fall * fabs(fx)) ? fmin(2.0 * t, 16.0) : 1.0;
      }
      const double f_old = fx;
#pragma unroll
      for (int e = 0; e < EPT; ++e) { u[e] = un[e]; g[e] = gn[e]; }
      fx = fn;
      over_acc = over_now; over_cnt_acc = over_cnt_now;
      f_slot ^= 1; f_valid = approx ? 0 : 1;
      if (!phaseS) ++it_full;
      if (sy > 1e-10 * sqrt(ss * yy)) {   // keep the pair (SPD update)
        int slot;
        if (k < st->maxcor) { slot = (head + k) % st->maxcor; ++k; }
        else { slot = head; head = (head + 1) % st->maxcor; }
#pragma unroll
        for (int e = 0; e < EPT; ++e)
          if (on[e]) { b.S[slot * ld + idx[e]] = sv[e]; b.Y[slot * ld + idx[e]] = yv[e]; }
        if (tid == 0) { b.rho[slot] = 1.0 / sy; b.yy[slot] = yy; }
        __threadfence_block();
        __syncthreads();   // the two-loop recursion below re-reads the pair from memory
      }
      ++it;
      const double fscale = fmax(fmax(fabs(f_old), fabs(fx)), 1.0);
      const double dec_now = f_old - fx;
      // Capped start (round 5 form).  The Ridge start regresses on the nearest-neighbour estimate and overshoots log-density x
      // volume, t = f + V, by up to e^14 in a few sparse cells of C3 and by e^28 where nearest-neighbour distances are noisy
      // in a high nominal dimension (tools/hard_cases.py: 500 passes of walking the exponential down one unit per step).
      // The objective kernels therefore continue e^t beyond t = cap by its second-order Taylor polynomial: a convex C^2
      // minorant whose curvature is bounded by e^cap -- cells above the cap sit on an exact quadratic, which one good
      // quasi-Newton step takes to its minimum -- and which IS the true objective (value, gradient, curvature) wherever no
      // row is above the cap.  The reduction reports whether any row was (b.over).  When the capped solve slows down with
      // rows still above the cap, the cap moves up by cap_step and the same point is evaluated again; a solve never ends
      // while the accepted point has rows above the cap (see `done` below).
      const bool slow = (f_old - fx) <= st->rebuild_tol * fscale && (f_old - fx) > st->ftol * fscale;
      if (cap < 1e300 && over_acc && it >= 3 && slow) { cap += st->cap_step; recap = true; }
      const bool revert = st->revert_after > 0 && st->it_at_resume >= 0 && !phaseS && it - st->it_at_resume > st->revert_after &&
                          (f_old - fx) > st->ftol * fscale;
      if (revert) {
        pause = true; pause_reason = 2;          // (solver.h: revert_after) the host restores the first preconditioner
      } else if (phaseS) {
        // the subsample objective has done its job once its own progress per iteration is small: same point, full objective
        // (not before a few iterations: the very first step from the Ridge start is a cautious t = 1 / |g|_1)
        if (it >= 4 && ((f_old - fx) <= st->sub_tol * fscale || n_eval_sub >= st->sub_max_evals)) { reeval = true; next_level = true; } else to_head = true;
      } else if (st->rebuild_armed > 0 && st->it_resume < 0 && !phase32 && !recap && it_full >= 2 && (slow || (it_full >= 4 && st->over_many > 0.0 && over_cnt_acc >= st->over_many && isfinite(fx)))) {
        pause = true;                    // the host rebuilds the preconditioner from the weights a = e^{f+V} at THIS point
      } else if (st->rebuild_armed > 0 && st->it_resume >= 0 && !approx && !recap && it - st->it_resume >= 6 && dec_prev > 0.0 &&
                 dec_prev2 > 0.0 && dec_now > 0.5 * dec_prev && dec_prev > 0.5 * dec_prev2 && dec_now > st->ftol * fscale) {
        // Round 5: ANOTHER rebuild.  Six or more iterations after the last one the loss still falls by less than half per
        // iteration, three times in a row: the weights have moved on from the point that preconditioner was built at
        // (diffusion-map-like coordinates, heavy tails: tools/hard_cases.py); on C3 the decrease contracts 50-100x per pass
        // after the first rebuild and this never fires.
        pause = true;
      } else if (phaseA && st->rebuild_armed && it_full >= 2 && (f_old - fx) <= st->rebuild_tol * fscale) {
        reeval = true;                   // mixed precision: the rebuild needs the rows' f -- anchor in fp64 HERE, then pause (below)
      } else if (phaseA) {
        // the plain 32-bit objective is a smooth surrogate whose optimum sits ~1e-9 (fixed point; fp32: ~5e-5) in
        // relative loss from the true one: once its progress per iteration falls below ftol32, evaluate in fp64 at
        // the same point and continue (corrected surrogate, or the fp64 buffer) with the pairs collected so far
        if ((f_old - fx) <= st->ftol32 * fscale) reeval = true; else to_head = true;
      } else if ((f_old - fx) <= st->ftol * fscale) {
        if (phaseC) reeval = true; else { status = 0; done = true; }   // phase C: the fp64 objective has the last word
      } else {
        const double dec = f_old - fx;
        bool small_gap = false;
        // (only in the asymptotic regime -- the decreases already tiny next to the loss: a loss that FALLS by orders of
        //  magnitude per pass, e^{f+V} coming down from an overshooting start, "contracts" too, and the estimate then
        //  declared convergence at a loss of 1e260: tools/robustness_diag_large.py)
        if (st->gap_tol > 0.0 && !phaseS && dec_prev > 0.0 && dec_prev2 > 0.0 && dec < 0.25 * dec_prev && dec_prev < 0.25 * dec_prev2 &&
            dec_prev2 <= 1e-4 * fscale) {
          const double r = dec / dec_prev;
          small_gap = dec * r / (1.0 - r) <= st->gap_tol * fscale;
        }
        if (small_gap) { if (phaseC) reeval = true; else { status = 0; done = true; } }
        else to_head = true;
      }
      if (!phaseS) { dec_prev2 = dec_prev; dec_prev = f_old - fx; }
    } else if (isfinite(fn) && fabs(fn - fx) <= st->ftol * fmax(fmax(fabs(fx), fabs(fn)), 1.0)) {
      // The trial changed the loss by no more than the stopping tolerance but was not a sufficient decrease: the
      // search has reached the rounding noise of the objective (sums of n terms), where shrinking the step further
      // only samples that noise -- seen as 5-7 wasted passes with t = 0.2, 0.02, ... before one happened to pass.
      // This is the relative-decrease test of the accepted branch applied to the rejected trial: converged.
      // (phase C with f_valid: the accepted point IS the last fp64 evaluation -- nothing has been accepted since -- so
      //  its loss, gradient and rows' f are already fp64: no second verification of the same point)
      if (approx && !(phaseC && f_valid)) { reeval = true; next_level = phaseS; } else { status = 0; done = true; }
    } else if (ls >= st->maxls) {
      if (approx && !(phaseC && f_valid)) { reeval = true; next_level = phaseS; }   // the surrogate is exhausted: continue in fp64 from the accepted point
      else { status = phaseC ? 0 : 2; done = true; }
    } else {
      if (isfinite(fn)) {
        const double tq = -gd * t * t / (2.0 * (fn - fx - gd * t));   // minimiser of the quadratic model
        t = fmin(fmax(tq, 0.1 * t), 0.5 * t);
      } else {
        t *= 0.1;
      }
#pragma unroll
      for (int e = 0; e < EPT; ++e) un[e] = fma(t, d[e], u[e]);
    }
  }
  if (to_head) {
    double gm = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) gm = fmax(gm, fabs(g[e]));
    gm = block_max(gm, red);
    if (!(gm > st->gtol)) {
      if (approx) { reeval = true; next_level = phaseS; } else { status = 0; done = true; }
    } else if (it >= st->maxiter) {
      status = 1; done = true;
    } else {
      // two-loop recursion over the k stored pairs (chronological j -> slot (head + j) % maxcor)
      double q[EPT];
#pragma unroll
      for (int e = 0; e < EPT; ++e) q[e] = g[e];
      // (the pair of the NEXT step is requested before the reduction of the current one: the L2 latency of the row
      //  loads -- ~1 us each -- would otherwise sit on the critical path 2 k times)
      double sv[EPT], yv[EPT], sn[EPT], yn[EPT];
      auto load_pair = [&](int j, double (&s_)[EPT], double (&y_)[EPT]) {
        const int slot = (head + j) % st->maxcor;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          s_[e] = on[e] ? b.S[slot * ld + idx[e]] : 0.0;
          y_[e] = on[e] ? b.Y[slot * ld + idx[e]] : 0.0;
        }
      };
      if (k > 0) load_pair(k - 1, sv, yv);
      for (int j = k - 1; j >= 0; --j) {
        const int slot = (head + j) % st->maxcor;
        if (j > 0) load_pair(j - 1, sn, yn); else load_pair(0, sn, yn);   // j == 0: first pair of the second loop
        double acc = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) acc = fma(sv[e], q[e], acc);
        const double a = b.rho[slot] * block_sum(acc, red);
        if (tid == 0) alpha_s[j] = a;
#pragma unroll
        for (int e = 0; e < EPT; ++e) { q[e] = fma(-a, yv[e], q[e]); sv[e] = sn[e]; yv[e] = yn[e]; }
      }
      if (k > 0) {
        const int last = (head + k - 1) % st->maxcor;
        const double gamma = 1.0 / (b.rho[last] * b.yy[last]);   // s.y / y.y of the newest pair
#pragma unroll
        for (int e = 0; e < EPT; ++e) q[e] *= gamma;
      }
      __syncthreads();
      for (int j = 0; j < k; ++j) {
        const int slot = (head + j) % st->maxcor;
        if (j + 1 < k) load_pair(j + 1, sn, yn);
        double acc = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) acc = fma(yv[e], q[e], acc);
        const double beta = b.rho[slot] * block_sum(acc, red);
        const double c = alpha_s[j] - beta;
#pragma unroll
        for (int e = 0; e < EPT; ++e) { q[e] = fma(sv[e], c, q[e]); sv[e] = sn[e]; yv[e] = yn[e]; }
      }
      double acc = 0.0;
#pragma unroll
      for (int e = 0; e < EPT; ++e) { d[e] = -q[e]; acc = fma(g[e], d[e], acc); }
      gd = block_sum(acc, red);
      if (!(gd < 0.0)) {   // not a descent direction (cannot happen for SPD pairs; guard anyway)
        k = 0; head = 0;
        acc = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) { d[e] = -g[e]; acc = fma(g[e], g[e], acc); }
        gd = -block_sum(acc, red);
      }
      // After phase C converged on F^ (relative decrease <= ftol, or its noise floor), this is the fp64 gradient at
      // that point and d the quasi-Newton step from it: the quadratic model promises a further decrease of |g.d| / 2.
      // If that is within the stopping tolerance, F itself has converged by the same measure the ftol test applies a
      // posteriori -- done, on fp64 evidence (the loss, the gradient and the rows' f of this very pass).
      if (verify && -0.5 * gd <= st->ftol * fmax(fabs(fx), 1.0)) { status = 0; done = true; }
      t = t0;
      if (after_switch && st->switch_t0 > 0.0) t = st->switch_t0;
      if (k == 0 && !resume) {
        double g1 = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) g1 += fabs(g[e]);
        g1 = block_sum(g1, red);
        t = fmin(1.0, 1.0 / g1);
      }
      ls = 0;
      mode = MLN_SOLVE_LS;
#pragma unroll
      for (int e = 0; e < EPT; ++e) un[e] = fma(t, d[e], u[e]);
    }
  }
  if (reeval) {           // same point, fp64 buffer: refreshes fx and g, keeps the curvature pairs
    // (a 32-bit surrogate left while still capped: everything from here on is the true objective; after the subsample
    //  phase the cap comes back -- the cells the subsample never saw are where e^{f+V} overshoots on the full objective)
    cap = phaseS ? st->cap0 : (phase32 ? __builtin_inf() : cap);
    // after a subsample: the next, larger one -- or, after the last, the full objective (on its 32-bit copy if there is one)
    if (phaseS && next_level && sub_level + 1 < st->n_sub_levels) { ++sub_level; gate = MLN_GATE_SUB; }
    else gate = phaseS ? st->gate_full : MLN_GATE_F64;
    mode = MLN_SOLVE_REEVAL;
#pragma unroll
    for (int e = 0; e < EPT; ++e) un[e] = u[e];
  }
  // a solve must not END on the capped objective: if the accepted point has rows above the cap, the cap moves up and the
  // point is evaluated again
  if (done && status == 0 && cap < 1e300 && over_acc && !approx) { done = false; cap += st->cap_step; recap = true; }
  if (recap && !reeval && !done) {   // the same point once more, on the uncapped objective (same copy)
    mode = MLN_SOLVE_REEVAL;
#pragma unroll
    for (int e = 0; e < EPT; ++e) un[e] = u[e];
  }
  if (pause) { if (tid == 0) st->gate_after_pause = gate; gate = MLN_GATE_PAUSE; }
  if (done) gate = MLN_GATE_DONE;
#pragma unroll
  for (int e = 0; e < EPT; ++e)
    if (on[e]) { b.u[idx[e]] = u[e]; b.g[idx[e]] = g[e]; b.un[idx[e]] = un[e]; b.d[idx[e]] = d[e]; }
  if (tid == 0) {
    st->gate = gate; st->mode = mode; st->status = status; st->it = it; st->n_eval = n_eval; st->n_eval32 = n_eval32;
    st->ls = ls; st->k = k; st->head = head; st->fx = fx; st->t = t; st->gd = gd;
    st->f_slot = f_slot; st->f_valid = f_valid;
    st->corr = corr; st->n_anchor = n_anchor; st->corr_k = corr_k; st->t0 = t0; st->cap = cap;
    st->n_eval_sub = n_eval_sub; st->it_full = it_full; st->sub_level = sub_level;
    st->dec_prev = dec_prev; st->dec_prev2 = dec_prev2;
    st->over_acc = over_acc; st->over_cnt_acc = over_cnt_acc;
    if (shrink) st->n_shrink += 1;
    if (pause) { st->rebuild_armed = (pause_reason == 1 && st->rebuild_armed > 0) ? st->rebuild_armed - 1 : 0; st->pause_reason = pause_reason; st->it_at_resume = -1; }
  }
}

__global__ void k_solver_init(SolverBuffers b, SolverState init, const double* __restrict__ u0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < init.m) {
    const double v = u0[i];
    b.u[i] = v; b.un[i] = v; b.g[i] = 0.0; b.gn[i] = 0.0; b.d[i] = 0.0;
  }
  if (i == 0) *b.st = init;
}

// after the host re-expressed the stored pairs in a new variable: s.y is invariant, y.y is not
__global__ __launch_bounds__(512) void k_solver_refresh_pairs(SolverBuffers b) {
  __shared__ double red[ST / 64];
  const SolverState* st = b.st;
  const int slot = blockIdx.x;
  if (slot >= st->maxcor) return;
  double acc = 0.0;
  for (int i = threadIdx.x; i < st->m; i += ST) { const double v = b.Y[slot * b.ld + i]; acc = fma(v, v, acc); }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) b.yy[slot] = acc;
}

__global__ void k_solver_resume(SolverBuffers b, int gate, int pairs_dropped, int revert_after, int rearm) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    SolverState* st = b.st;
    st->gate = gate;
    st->mode = MLN_SOLVE_RESUME;
    if (pairs_dropped) { st->k = 0; st->head = 0; }
    st->resume_keep_pairs = pairs_dropped ? 0 : 1;
    st->revert_after = revert_after;
    st->it_at_resume = revert_after > 0 ? st->it : -1;
    st->it_resume = st->it;       // (the next rebuild waits for slow linear convergence at least six iterations from here)
    if (rearm >= 0) st->rebuild_armed = rearm;
    st->pause_reason = 0;
  }
}

__global__ void k_solver_resume_plain32(SolverBuffers b, int gate, int m, int pairs_dropped) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) b.un[i] = b.u[i];
  if (i == 0) {
    SolverState* st = b.st;
    st->gate = gate;
    st->mode = MLN_SOLVE_REEVAL;
    if (pairs_dropped) { st->k = 0; st->head = 0; }
    st->corr = 0; st->n_anchor = 0; st->corr_k = 0.0;
    st->revert_after = 0; st->it_at_resume = -1; st->pause_reason = 0;
    st->rebuild_armed = 0; st->it_resume = st->it;      // (the mixed solve's rebuild is one attempt, as it always was)
  }
}

}  // namespace

int launch_solver_resume_plain32(mln_ctx* ctx, const SolverBuffers& b, int gate, int m, int pairs_dropped) {
  hipLaunchKernelGGL(k_solver_resume_plain32, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, b, gate, m, pairs_dropped);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_solver_refresh_pairs(mln_ctx* ctx, const SolverBuffers& b, int maxcor) {
  hipLaunchKernelGGL(k_solver_refresh_pairs, dim3((unsigned)maxcor), dim3(ST), 0, ctx->stream, b);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_solver_resume(mln_ctx* ctx, const SolverBuffers& b, int gate, int pairs_dropped, int revert_after, int rearm) {
  hipLaunchKernelGGL(k_solver_resume, dim3(1), dim3(64), 0, ctx->stream, b, gate, pairs_dropped, revert_after, rearm);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_solver_init(mln_ctx* ctx, const SolverBuffers& b, const SolverState& init, const double* u0) {
  hipLaunchKernelGGL(k_solver_init, dim3((unsigned)((init.m + 255) / 256)), dim3(256), 0, ctx->stream, b, init, u0);
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}

int launch_solver_step(mln_ctx* ctx, const SolverBuffers& b, int m) {
  const int ept = (m + ST - 1) / ST;
  const dim3 grid(1), block(ST);
#define MLN_STEP(E) case E: hipLaunchKernelGGL(k_solver_step<E>, grid, block, 0, ctx->stream, b); break;
  switch (ept) {
    MLN_STEP(1) MLN_STEP(2) MLN_STEP(3) MLN_STEP(4) MLN_STEP(5) MLN_STEP(6) MLN_STEP(7) MLN_STEP(8)
    MLN_STEP(9) MLN_STEP(10) MLN_STEP(11) MLN_STEP(12) MLN_STEP(13) MLN_STEP(14) MLN_STEP(15) MLN_STEP(16)
    default: mln_set_error(ctx, "solver: m > 8192 is not supported by this build"); return MLN_ERR_UNSUPPORTED;
  }
#undef MLN_STEP
  MLN_HIP(ctx, hipGetLastError());
  return MLN_OK;
}
