"""Pass count of the C3 fit when the sampled rows of K entering the preconditioner's Gram are rounded to b fractional
bits (MELLON_AMD_GRAM_QBITS): how exact must K_s^T K_s be, given that Lp^-1 . Lp^-T amplifies its error?"""
import os as _os; _os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")   # this tool turns experiment knobs (csrc/mln_options.h)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, mellon_amd
from mellon_amd import _lib
ctx = _lib.default_context()
n, d, m = 1_000_000, 50, 5000
for seed in [int(s) for s in os.environ.get("SEEDS", "3").split(",")]:
    x = bench.gaussian_mixture(n, d, seed)
    lm, _ = bench.make_landmarks(x, m, "device", ctx)
    xd = ctx.to_device(x)
    nn = ctx.nn_distances(xd, xd)
    ref = None
    for bits in os.environ.get("BITS", "0,40,32,28,24,20,16").split(","):
        os.environ["MELLON_AMD_GRAM_QBITS"] = bits
        est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
        t0 = time.perf_counter()
        dens = est.fit_predict(xd)
        dt = time.perf_counter() - t0
        st = est._fit.stage_times()
        est._fit.close()
        if ref is None:
            ref = dens.copy()
        print(seed, "bits", bits, {"step_ms": round(1e3 * dt, 1), "n32": st["objective32_launches"], "n64": st["objective_launches"],
                                   "rel": float(np.abs(dens - ref).max() / np.abs(ref).max())}, flush=True)
