"""Per-evaluation trace of the default (pure fp64) MAP solve at C3: loss, step length, mode (0 first, 1 line-search trial,
2 re-evaluation, 3 resume), gate (which copy / subsample the pass streamed).  MELLON_AMD_TRACE=2 prints it from the library."""
import os, sys
sys.path.insert(0, ".")
os.environ["MELLON_AMD_MIXED"] = "0"
import numpy as np, bench, mellon_amd
from mellon_amd import _lib
ctx = _lib.default_context()
n, d, m = 1_000_000, 50, 5000
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
x = bench.gaussian_mixture(n, d, seed)
xd = ctx.to_device(x)
lm = np.ascontiguousarray(ctx.kmeans(x[:100_000], m, seed=42).astype(np.float32).astype(np.float64))
nn = ctx.nn_distances(xd)
est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
est.prepare_inference(xd)
os.environ["MELLON_AMD_TRACE"] = "2"
z, l, ne, ni, st = est._fit.map_solve(est.initial_value)
print("evals", ne, "iters", ni, est._fit.stage_times())
