"""Evaluations and step time of the C3 fit against the L-BFGS memory (maxcor)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, mellon_amd
from mellon_amd import _lib
ctx = _lib.default_context()
n, d, m = 1_000_000, 50, 5000
x = bench.gaussian_mixture(n, d, 3)
out = {}
for how in ("device",):
    lm, _ = bench.make_landmarks(x, m, how, ctx)
    xd = ctx.to_device(x)
    nn = ctx.nn_distances(xd, xd)
    for mc in [int(a) for a in sys.argv[1:]] or [30, 20, 15, 10, 7]:
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
            est.lbfgsb_options = dict(maxcor=mc)
            dens = est.fit_predict(xd)
            dt = time.perf_counter() - t0
            st = est._fit.stage_times()
            ev = est.loss_func.n_eval
            est._fit.close()
            if rep > 0 and (best is None or dt < best[0]):
                best = (dt, ev, st["objective32_launches"], st["objective_launches"], dens.copy())
        if mc == 30:
            ref = best[4]
        out[mc] = {"step_ms": 1e3 * best[0], "evals": best[1], "fp32": best[2], "fp64": best[3],
                   "rel_vs_maxcor30": float(np.abs(best[4] - ref).max() / np.abs(ref).max())}
        print(mc, out[mc], flush=True)
