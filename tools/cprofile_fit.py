import os, sys, cProfile, pstats, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, mellon_amd
from mellon_amd import _lib
ctx = _lib.default_context()
n, d, m = 1_000_000, 50, 5000
x = bench.gaussian_mixture(n, d, 3); lm = bench.make_landmarks(x, m); xd = ctx.to_device(x); nn = ctx.nn_distances(xd)
def run():
    est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn)
    return est.fit_predict(xd)
run(); gc.collect()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
