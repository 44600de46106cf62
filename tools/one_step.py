"""Three fp64 C3 steps with device landmarks (for kernel traces)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELLON_AMD_MIXED", "0")
import bench, mellon_amd
from mellon_amd import _lib
ctx = _lib.default_context()
x = bench.gaussian_mixture(1_000_000, 50, 3)
lm, _ = bench.make_landmarks(x, 5000, "device", ctx)
xd = ctx.to_device(x)
nn = ctx.nn_distances(xd, xd)
for rep in range(3):
    est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
    est.fit_predict(xd)
    est._fit.close()
