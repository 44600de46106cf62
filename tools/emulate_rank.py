"""What ONE rank of an N-rank strong-scaling run of C3 computes, timed on one GPU: rank 0's shard (1e6 / N cells), the
Gram sample stride of the GLOBAL problem, everything replicated (landmark factorisations, preconditioner, the
optimiser) in full.  The collectives are absent (one process): their cost is modelled separately in DESIGN.md S5.

From 3 ranks on the m x m right-hand-side work (whitening of the Gram, C^-1, P) is split by columns over the ranks:
MELLON_AMD_EMULATE_RANKS=N makes the library time rank 0's block and compute the other blocks untimed (their wall time
comes back as stage_times()["emulation_excluded_s"] and is subtracted here).

    python tools/emulate_rank.py 1 2 4 8 > profiles/rNN_emulated_ranks.json
"""
import os as _os; _os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")   # this tool turns experiment knobs (csrc/mln_options.h)
import gc, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, mellon_amd
from mellon_amd import _lib, distributed


class OneOfN(distributed.Communicator):
    def __init__(self, n_ranks):
        self.n_ranks = n_ranks

    def global_count(self, n_local):
        return int(n_local) * self.n_ranks

    def global_offset(self, n_local):
        return 0, int(n_local) * self.n_ranks


if os.environ.get("MIXED", "0") == "0":
    os.environ["MELLON_AMD_MIXED"] = "0"          # the headline mode: float64 throughout
ctx = _lib.default_context()
n, d, m = 1_000_000, 50, 5000
x = bench.gaussian_mixture(n, d, 3)
lm, _ = bench.make_landmarks(x, m, "device", ctx)
xd = ctx.to_device(x)
nn = ctx.nn_distances(xd, xd)
out = {}
for N in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    distributed.set_current(OneOfN(N))
    os.environ["MELLON_AMD_EMULATE_RANKS"] = str(N)
    lo, hi = distributed.shard_bounds(n, N, 0)
    xs = ctx.to_device(x[lo:hi])
    best = None
    for rep in range(4):
        t0 = time.perf_counter()
        est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn[lo:hi], check_rank=False)
        dens = est.fit_predict(xs)
        dt = time.perf_counter() - t0
        st = est._fit.stage_times()
        dt -= st.get("emulation_excluded_s", 0.0)
        ev = est.loss_func.n_eval
        est._fit.close()
        del est
        if rep > 0 and (best is None or dt < best[0]):
            best = (dt, st, ev)
    dt, st, ev = best
    n32, n64 = st["objective32_launches"], st["objective_launches"]
    out[str(N)] = {"cells_on_this_rank": hi - lo, "step_ms": 1e3 * dt, "evaluations": ev,
                   "objective_kernels_ms": 1e3 * (st["objective_kernel_s"] + st["objective32_kernel_s"]),
                   "fp32_pass_ms": 1e3 * st["objective32_kernel_s"] / max(n32, 1), "fp64_pass_ms": 1e3 * st["objective_kernel_s"] / max(n64, 1),
                   "kernel_matrix_ms": 1e3 * st["kernel_matrix_s"], "chol_Lp_ms": 1e3 * st["cholesky_s"],
                   "gram_and_solves_ms": 1e3 * st["ridge_gram_s"], "chol_C_inverses_ms": 1e3 * st["ridge_solve_s"],
                   "sub_passes_ms": 1e3 * st["objective_sub_kernel_s"], "sub_evaluations": st["objective_sub_launches"],
                   "rebuild_ms": 1e3 * st["precond_rebuild_s"], "rebuilds": st["precond_rebuilds"],
                   "full_pass_equivalents": st["objective_pass_equivalents"], "fp64_launches": n64, "launches_32bit": n32}
    xs.free()
    gc.collect()
distributed.set_current(distributed.Communicator())
one = out.get("1", {}).get("step_ms")
for k, v in out.items():
    v["speedup_without_communication"] = None if not one else one / v["step_ms"]
# COMPOSED estimate.  The emulation solves rank 0's shard as if it were the whole problem, so its evaluation COUNT is that of
# an n / N-cell problem (and its rebuilt preconditioner comes from this rank's share of the sample alone).  A real N-rank
# run follows the trajectory of the whole problem -- the sums are the same numbers however many ranks add them -- i.e. the
# counts of the N = 1 line.  Per-launch costs from this rank's emulation, counts from N = 1:
#   step(N) = [step(N) - objective kernels(N) - rebuild(N)]        replicated set-up, per-evaluation launches, host
#             * evaluations(1) / evaluations(N) on the per-evaluation part (0.14 ms each)
#             + full passes(1) * fp64 pass(N) + sub passes(1) * sub pass(N) + rebuild(N) if the N = 1 solve rebuilt
ref = out.get("1")
if ref:
    for k, v in out.items():
        per_eval_ms = 0.14
        other = v["step_ms"] - v["objective_kernels_ms"] - v["sub_passes_ms"] - v["rebuild_ms"]
        other += per_eval_ms * (ref["evaluations"] - v["evaluations"])
        sub_ms = v["sub_passes_ms"] / max(v["sub_evaluations"], 1)
        reb = v["rebuild_ms"] if v["rebuilds"] else None
        comp = other + ref["fp64_launches"] * v["fp64_pass_ms"] + ref["sub_evaluations"] * sub_ms
        v["composed_global_trajectory_ms"] = None if (ref["rebuilds"] and reb is None) else comp + (reb or 0.0)
        v["composed_note"] = ("N = 1 counts (%d full + %d sub passes, %d rebuild) at this rank's per-launch costs"
                              % (ref["fp64_launches"], ref["sub_evaluations"], ref["rebuilds"])) + \
                             ("" if reb is not None or not ref["rebuilds"] else "; this emulation did not rebuild: run with MELLON_AMD_REBUILD=1")
        if v["composed_global_trajectory_ms"]:
            v["composed_speedup_without_communication"] = ref["step_ms"] / v["composed_global_trajectory_ms"]
print(json.dumps(out, indent=1))
