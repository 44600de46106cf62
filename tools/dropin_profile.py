"""cProfile of the drop-in call DensityEstimator().fit_predict(X) on 1e6 x 50 HOST cells (nothing precomputed): the steps
before the path (k-means landmarks, 1-NN distances) and the fit.   python tools/dropin_profile.py [n]"""
import os, sys, cProfile, pstats, gc, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, mellon_amd
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
x = bench.gaussian_mixture(n, 50, 3)
def run():
    est = mellon_amd.DensityEstimator(check_rank=False if os.environ.get("DROPIN_NO_RANK") else None)
    out = est.fit_predict(x)
    est._fit.close()
    return out
for i in range(3):
    gc.collect(); t0 = time.perf_counter(); run(); print(f"call {i}: {time.perf_counter() - t0:.2f} s", flush=True)
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
