"""Pass counts of the C3 fit when the 32-bit fixed-point copy is rounded to b bits (MELLON_AMD_COPY_BITS) and the
switch tolerance of the uncorrected phase scales with the copy's error: how narrow a copy does the corrected solve bear?"""
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
    for bits in [32, 28, 24, 20, 16]:
        for mult in (8.0, 64.0):
            if bits == 32 and mult != 8.0:
                continue
            os.environ.pop("MELLON_AMD_COPY_BITS", None)
            if bits < 32:
                os.environ["MELLON_AMD_COPY_BITS"] = str(bits)
            os.environ["MELLON_AMD_MIXED_FTOL"] = repr(mult * 2.0 ** -(bits + 1))
            est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
            dens = est.fit_predict(xd)
            st = est._fit.stage_times()
            est._fit.close()
            if ref is None:
                ref = dens.copy()
            print(seed, "bits", bits, "ftolA", os.environ["MELLON_AMD_MIXED_FTOL"], {"n32": st["objective32_launches"], "n64": st["objective_launches"],
                  "rel": float(np.abs(dens - ref).max() / np.abs(ref).max())}, flush=True)
