"""cProfile of the plain drop-in call (second call in the process: the first pays context creation and allocator warm-up)."""
import cProfile, pstats, sys, time
import numpy as np
sys.path.insert(0, ".")
import bench, mellon_amd
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 50
X = bench.gaussian_mixture(n, d, 3)
for rep in range(2):
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    est = mellon_amd.DensityEstimator()
    dens = est.fit_predict(X)
    pr.disable()
    print(f"call {rep}: {time.perf_counter() - t0:.2f} s")
    est._fit.close()
ps = pstats.Stats(pr)
ps.sort_stats("cumulative").print_stats(45)
