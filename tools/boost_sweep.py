"""Pass counts of the C3 fit against the slope ratio above which the first trial step of the next line search doubles
(solver.hip "step-length memory"; 0 = every search starts at t = 1)."""
import os as _os; _os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")   # this tool turns experiment knobs (csrc/mln_options.h)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, mellon_amd
from mellon_amd import _lib
ctx = _lib.default_context()
n, d, m = 1_000_000, 50, 5000
for seed in [int(s) for s in os.environ.get("SEEDS", "3,7").split(",")]:
    x = bench.gaussian_mixture(n, d, seed)
    lm, _ = bench.make_landmarks(x, m, "device", ctx)
    xd = ctx.to_device(x)
    nn = ctx.nn_distances(xd, xd)
    ref = None
    for b in os.environ.get("BOOSTS", "0,0.5,0.3,0.2,0.1").split(","):
        os.environ["MELLON_AMD_LS_BOOST"] = b.split("/")[0]
        os.environ["MELLON_AMD_LS_BOOST_FALL"] = b.split("/")[1] if "/" in b else "0.25"
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
            dens = est.fit_predict(xd)
            dt = time.perf_counter() - t0
            st = est._fit.stage_times()
            est._fit.close()
            if best is None or dt < best[0]:
                best = (dt, st, dens.copy())
        if ref is None:
            ref = best[2]
        st = best[1]
        print(seed, "boost", b, {"step_ms": round(1e3 * best[0], 1), "n32": st["objective32_launches"], "n64": st["objective_launches"],
                                 "km_ms": round(1e3 * st["kernel_matrix_s"], 1),
                                 "rel": float(np.abs(best[2] - ref).max() / np.abs(ref).max())}, flush=True)
