"""Pass counts of the C3 fit against the cap beyond which e^{f+V} is continued linearly during the steep first part of the
solve (solver.hip "capped start"; "off" = never capped; "c/f": cap c, dropped when the loss falls by less than f per pass)."""
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
    for cap in os.environ.get("CAPS", "off,2,4,6,8,12").split(","):
        os.environ["MELLON_AMD_EXP_CAP"] = "off"
        if cap != "off":
            os.environ["MELLON_AMD_EXP_CAP"] = cap.split("/")[0]
            os.environ["MELLON_AMD_EXP_CAP_FALL"] = cap.split("/")[1] if "/" in cap else "0.15"
        est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
        t0 = time.perf_counter()
        dens = est.fit_predict(xd)
        dt = time.perf_counter() - t0
        st = est._fit.stage_times()
        est._fit.close()
        if ref is None:
            ref = dens.copy()
        print(seed, "cap", cap, {"step_ms": round(1e3 * dt, 1), "n32": st["objective32_launches"], "n64": st["objective_launches"],
                                 "rel": float(np.abs(dens - ref).max() / np.abs(ref).max())}, flush=True)
