"""C3 fits under different switches of the MAP solve's iteration path (subsample start, preconditioner rebuild, their
tolerances), pure fp64 unless MIXED=1: pass counts in full-pass equivalents, step time, distance from the first variant.

    VARIANTS="SUBSAMPLE=0,REBUILD=0;SUBSAMPLE=1,REBUILD=0;SUBSAMPLE=1,REBUILD=1" SEEDS=3,7 python tools/solver_sweep.py
(each variant: comma-separated NAME=value pairs, set as MELLON_AMD_NAME)."""
import os as _os; _os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")   # this tool turns experiment knobs (csrc/mln_options.h)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, mellon_amd
from mellon_amd import _lib
ctx = _lib.default_context()
n, d, m = int(os.environ.get("N", 1_000_000)), int(os.environ.get("D", 50)), int(os.environ.get("M", 5000))
kern = getattr(mellon_amd.cov, os.environ.get("KERNEL", "Matern52"))
if os.environ.get("MIXED", "0") == "0":
    os.environ["MELLON_AMD_MIXED"] = "0"
variants = os.environ.get("VARIANTS", "SUBSAMPLE=0,REBUILD=0;SUBSAMPLE=1,REBUILD=0;SUBSAMPLE=0,REBUILD=1;SUBSAMPLE=1,REBUILD=1").split(";")
for seed in [int(s) for s in os.environ.get("SEEDS", "3").split(",")]:
    x = bench.gaussian_mixture(n, d, seed)
    lm, _ = bench.make_landmarks(x, m, "device", ctx)
    xd = ctx.to_device(x)
    nn = ctx.nn_distances(xd, xd)
    ref = None
    for var in variants:
        keys = []
        for kv in [v for v in var.split(",") if v]:
            k, v = kv.split("=")
            os.environ["MELLON_AMD_" + k] = v
            keys.append("MELLON_AMD_" + k)
        best = None
        for rep in range(int(os.environ.get("REPS", 2))):
            est = mellon_amd.DensityEstimator(cov_func_curry=kern, landmarks=lm, nn_distances=nn, check_rank=False)
            t0 = time.perf_counter()
            dens = est.fit_predict(xd)
            dt = time.perf_counter() - t0
            st = est._fit.stage_times()
            ev = est.loss_func.n_eval
            est._fit.close()
            if best is None or dt < best[0]:
                best = (dt, st, ev)
        dt, st, ev = best
        if ref is None:
            ref = dens.copy()
        print(seed, var, {"step_ms": round(1e3 * dt, 1), "evals": ev, "n64": st["objective_launches"], "n32": st["objective32_launches"],
                          "nsub": st["objective_sub_launches"], "stride": st["objective_sub_stride"],
                          "pass_equiv": round(st["objective_pass_equivalents"], 2), "rebuild_ms": round(1e3 * st["precond_rebuild_s"], 1),
                          "rebuilds": st["precond_rebuilds"], "obj64_ms": round(1e3 * st["objective_kernel_s"], 1),
                          "sub_ms": round(1e3 * st["objective_sub_kernel_s"], 1),
                          "gram_ms": round(1e3 * st["ridge_gram_s"], 1), "factor_ms": round(1e3 * st["ridge_solve_s"], 1),
                          "rel": float(np.abs(dens - ref).max() / np.abs(ref).max())}, flush=True)
        for k in keys:
            del os.environ[k]
