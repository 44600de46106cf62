import os, sys
sys.path.insert(0, ".")
os.environ["MELLON_AMD_EXPERIMENTAL"] = "1"
os.environ["MELLON_AMD_TRACE"] = "2"
import numpy as np, mellon_amd
from mellon_amd import _lib
ctx = _lib.default_context()
rng = np.random.default_rng(11)
n, m = 1_000_000, 2000
# same stream of random numbers as tools/robustness_diag_large.py's t3 case run alone
x = np.ascontiguousarray(rng.standard_t(3, size=(n, 20)))
xd = ctx.to_device(x); nn = ctx.nn_distances(xd, xd); lm = ctx.kmeans(x[:100000], m, seed=42)
est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
dens = est.fit_predict(xd)
print("status", est.opt_state, "dens range", dens.min(), dens.max(), est._fit.stage_times())
