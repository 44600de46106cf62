// Device-resident L-BFGS state of the MAP solve (solver.hip).  Reference: inference.minimize_lbfgsb
// (inference.py:272-288) = SciPy L-BFGS-B without bounds; here the optimiser's state and every decision
// live on the GPU, so that between two passes over the n x m buffer there is no host round trip.
#pragma once
#include "mln_core.h"

enum { MLN_SOLVE_FIRST = 0, MLN_SOLVE_LS = 1, MLN_SOLVE_REEVAL = 2, MLN_SOLVE_RESUME = 3 };

struct SolverState {
  int gate;      // which copy of the buffer the next evaluation streams: MLN_GATE_F64, MLN_GATE_F32 (the plain 32-bit
                 // surrogate), MLN_GATE_F32C (the 32-bit surrogate with its first-order correction); MLN_GATE_DONE
  int mode;      // what the evaluation in flight is: first point, line-search trial, re-evaluation at u (phase switch)
  int status;    // 0 converged, 1 maxiter, 2 line search failed
  int it, n_eval, n_eval32, ls;
  int k, head;   // curvature pairs stored / slot of the oldest
  int maxiter, maxcor, maxls;
  int m;
  int f_slot;    // which of the two per-row f buffers holds f at the ACCEPTED point (the pass in flight writes the other)
  int f_valid;   // ... and whether that evaluation streamed the fp64 buffer
  int corr;      // a first-order correction (c, corr_k) of the 32-bit surrogate is in force
  int n_anchor;  // fp64 evaluations that (re)anchored it
  int use_corr;  // set by the host: continue on the corrected surrogate after the first fp64 evaluation
  double ftol, gtol, ftol32;
  double prior_const;       // (m / 2) log 2 pi
  double fx, t, gd;         // accepted loss, current trial step, g . d at the accepted point
  double cap;               // the likelihood's e^t is continued by its second-order Taylor polynomial beyond t = cap (inf: off);
                            // raised by cap_step whenever the capped solve slows down with rows still above it
  double cap_step;
  int over_acc;             // the ACCEPTED point was evaluated with rows above the cap (its loss / gradient are the capped objective's)
  double cap0;              // its initial value: the cap comes back when the subsample phase hands over to the full objective
                            // (the cells the subsample never saw are where e^{f+V} overshoots then)
  double boost_fall;        // ... and the relative decrease of the loss per pass above which that rule applies
  double t0, boost;         // first trial step of the next line search; slope ratio above which it doubles (0: always 1)
  double corr_k;            // corrected surrogate: F^(u) = F32(u) + c . u + corr_k,  grad F^ = grad F32 + c
  // Row-subsample start (gate == MLN_GATE_SUB): the solve begins on the MAP problem of every row_stride-th cell (the
  // cells of the preconditioner's Gram) -- passes at 1 / row_stride of the bytes -- and moves to the full objective
  // (gate_full) at the same point once its progress per iteration falls below sub_tol.
  double sub_tol;
  int sub_max_evals;        // ... or after this many evaluations on the subsample (a slow subsample problem is not worth finishing)
  int gate_full;            // MLN_GATE_F64, or MLN_GATE_F32 when a 32-bit copy exists
  int n_eval_sub;           // evaluations on the subsamples so far
  int sub_level, n_sub_levels;   // ... which of the (nested, ever larger) subsamples the SUB gate currently means
  // Preconditioner rebuild: with rebuild_armed set by the host, the solver PAUSES (gate = MLN_GATE_PAUSE) after an
  // accepted fp64 iteration whose progress has fallen below rebuild_tol; the host then re-factors the preconditioner
  // from the a-weighted importance sample at that point (api.hip fit_rebuild_precond), re-expresses u and g in the new
  // variable and resumes (mode = MLN_SOLVE_RESUME, no pairs).
  int rebuild_armed;        // rebuilds the host still allows (0: none; each pause for a rebuild takes one)
  int it_resume;            // iteration count at the last resume after a rebuild (-1: none yet)
  double over_many;         // rows above the cap (all ranks) from which the start counts as overshooting: the FIRST rebuild then
                            // does not wait for slow progress (0: never early)
  double over_cnt_acc;      // rows above the cap at the accepted point
  int it_full;              // accepted iterations on the full objective
  int resume_keep_pairs;    // MLN_SOLVE_RESUME: the curvature pairs are still valid (no new variable)
  int gate_after_pause;     // the copy the solve continues on once the host resumes it
  double rebuild_tol;
  // First trial step of the first line search on the full objective after the subsample phase.  The quasi-Newton step
  // built on the subsample's curvature is too long there -- the cells the subsample never saw still carry large weights
  // e^{f+V} -- and t = 1 was rejected on most data sets tried (a wasted pass); the accepted lengths were 0.30-0.39.
  double switch_t0;
  // Estimated remaining gap.  Once the decrease per iteration contracts by more than 4x twice in a row (r = dec_k /
  // dec_{k-1} < 1/4: the regime after the preconditioner rebuild, where it contracts 50-100x per pass) the loss still to
  // gain is ~ dec_k r / (1 - r); the solve stops when that is below gap_tol * max(|f|, 1) instead of spending one or two
  // more passes to watch the decrease itself fall below ftol.  0: off (SciPy's ftol test alone).
  double gap_tol;
  double dec_prev, dec_prev2;   // decrease of the loss in the last two accepted iterations on the full objective (0: none yet)
  // Pathological start (round 4, tools/robustness_sweep_large.py).  The Ridge start regresses on the nearest-neighbour
  // estimate and may overshoot log-density x volume by HUNDREDS where nearest-neighbour distances span many decades in a
  // high nominal dimension (1e6 cells of a 3-D tree embedded in 20-D: e^{f+V} overflows; the loss is 1e260 or inf).
  // From there a quasi-Newton step walks the exponential down one unit per pass, and every relative stopping test is
  // meaningless.  While the FIRST evaluation is not finite or above start_cap, the start is halved (z = C^-T u is linear
  // in u; z = 0 is the constant density mu, always finite) and evaluated again -- at most 64 times.
  double start_cap;
  int n_shrink;
  // Second preconditioner on trial: it is built for the end game (a handful of Newton-like iterations).  If the solve has
  // not converged revert_after accepted iterations after the resume, that preconditioner is not what it was built to be
  // (weights e^{f+V} spanning too many decades at the pause point for the importance sample): the solver pauses once
  // more with pause_reason = 2 and the host puts the first preconditioner back.  0: off.
  int revert_after;
  int it_at_resume;         // -1: not armed
  int pause_reason;         // 1 rebuild, 2 revert
};

struct SolverBuffers {
  SolverState* st;
  double *u, *g, *un, *gn, *d;   // m each: accepted point / gradient, trial point / gradient, direction
  double *S, *Y;                 // maxcor x ld
  double *rho, *yy;              // maxcor each: 1 / s.y and y.y
  double* c;                     // m: gradient of (fp64 objective - 32-bit surrogate) at the last anchor
  const double* z;               // z = C^-T un of the evaluation in flight (m): the prior is 1/2 |z|^2 ...
  const double* z2;              // ... or, when given, 1/2 z . z2  (implicit mode: z = w, z2 = Kj w: 1/2 w^T Kj w = 1/2 |Lp^T w|^2)
  const double* lik;             // its (all-reduced) likelihood sum
  const double* over;            // > 0: some row of that evaluation lay above the cap (capped and true objective differ there)
  int64_t ld;
  double* trace;                 // optional: 4 doubles per evaluation (loss, step, mode, gate), 512 entries
};

// u0 (device, m) is copied into u and un; the first evaluation is then enqueued by the caller
int launch_solver_init(mln_ctx* ctx, const SolverBuffers& b, const SolverState& init, const double* u0);
// consumes the evaluation in flight (gn, z, lik) and prepares the next trial point in un -- or sets gate = DONE
int launch_solver_step(mln_ctx* ctx, const SolverBuffers& b, int m);
// after a pause: the host has written the accepted point and its gradient in the (new) preconditioned variable into
// b.u / b.g; the next step starts a line search from there on `gate` (pairs_dropped: the history starts over)
// rearm >= 0: the number of rebuilds still allowed becomes that (a rebuild that declined twice is not asked for again)
int launch_solver_resume(mln_ctx* ctx, const SolverBuffers& b, int gate, int pairs_dropped, int revert_after = 0, int rearm = -1);
// after a pause at the mixed solve's early fp64 anchor whose rebuild was declined: forget the anchor (it was taken far from
// the optimum, where the 32-bit surrogate and the fp64 objective differ by more than a first-order correction mends) and
// go on with the PLAIN 32-bit surrogate from the same point -- one re-evaluation there on `gate`, history kept (or dropped:
// after a revert of the preconditioner the variable is a new one)
int launch_solver_resume_plain32(mln_ctx* ctx, const SolverBuffers& b, int gate, int m, int pairs_dropped = 0);
// the stored pairs were re-expressed in a new variable by the host (S <- T S, Y <- T^-T Y): recompute y.y per slot
int launch_solver_refresh_pairs(mln_ctx* ctx, const SolverBuffers& b, int maxcor);
