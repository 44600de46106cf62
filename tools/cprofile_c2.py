"""cProfile of one C2 fit_predict (1e5 x 20, m = 1000, ExpQuad) from DEVICE cells: where the host-side milliseconds of a small
step go.   python tools/cprofile_c2.py"""
import os, sys, cProfile, pstats, gc, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELLON_AMD_MIXED", "0")
import bench, mellon_amd
from mellon_amd import _lib
ctx = _lib.default_context()
n, d, m = 100_000, 20, 1000
x = bench.gaussian_mixture(n, d, 2); lm, _ = bench.make_landmarks(x, m, "device", ctx); xd = ctx.to_device(x); nn = ctx.nn_distances(xd)
def run(xin):
    est = mellon_amd.DensityEstimator(cov_func_curry=mellon_amd.cov.ExpQuad, landmarks=lm, nn_distances=nn, check_rank=False)
    out = est.fit_predict(xin)
    est._fit.close()
    return out
for _ in range(3): run(xd)
gc.collect()
t0 = time.perf_counter()
for _ in range(10): run(xd)
print("step ms", 1e2 * (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): run(xd)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(35)
