"""Evaluation counts of the MAP solve over different landmark sets (mixed vs pure fp64)."""
import os as _os; _os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")   # this tool turns experiment knobs (csrc/mln_options.h)
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, bench, mellon_amd
from mellon_amd import _lib
ctx = _lib.default_context()
n, d, m = 1_000_000, 50, 5000
x = bench.gaussian_mixture(n, d, 3); xd = ctx.to_device(x); nn = ctx.nn_distances(xd)
for seed in (42, 1, 2, 3, 4, 5):
    lm = bench.make_landmarks(x, m, seed=seed)
    row = []
    for mixed in ("1", "0"):
        os.environ["MELLON_AMD_MIXED"] = mixed
        est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn)
        t0 = time.perf_counter(); dens = est.fit_predict(xd); dt = time.perf_counter() - t0
        st = est._fit.stage_times()
        row.append((est.loss_func.n_eval, int(st["objective32_launches"]), int(st["objective_launches"]), round(dt * 1e3)))
        est._fit.close()
    print("seed", seed, "mixed (evals, fp32, fp64, ms):", row[0], " fp64-only:", row[1], flush=True)
