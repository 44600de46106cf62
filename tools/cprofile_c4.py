"""cProfile of the C4 step (TimeSensitiveDensityEstimator.fit_predict, 5e5 x 30 + time column, 8 time points, m = 2000) as
bench.py --config c4 runs it (host cells): where the host-side milliseconds go.   python tools/cprofile_c4.py"""
import os, sys, cProfile, pstats, gc, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELLON_AMD_MIXED", "0")
import numpy as np
import bench, mellon_amd
from mellon_amd import _lib
from mellon_amd.parameters import compute_nn_distances_within_time_points
ctx = _lib.default_context()
n, d, m, T, ls_time = 500_000, 30, 2000, 8, 1.5
xt = bench.c4_workload(n, d, T, 4)
nn = compute_nn_distances_within_time_points(xt, local=(0, n))
ls = float(np.exp(np.mean(np.log(nn)) + 3.0))
rng = np.random.default_rng(4)
km = xt[rng.choice(n, 20000, replace=False)].copy(); km[:, -1] *= ls / ls_time
lm = ctx.kmeans(km, m, seed=42); lm[:, -1] /= ls / ls_time
lm = np.ascontiguousarray(lm.astype(np.float32).astype(np.float64))
kern = mellon_amd.cov.Matern52
def run():
    est = mellon_amd.TimeSensitiveDensityEstimator(cov_func_curry=kern, landmarks=lm, nn_distances=nn, ls_time=ls_time, d=d, check_rank=False)
    return est, est.fit_predict(xt)
for _ in range(3): run()
gc.collect()
t0 = time.perf_counter()
for _ in range(5): run()
print("step ms", 2e2 * (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): run()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
