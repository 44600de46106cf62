"""The plain drop-in call at scale: DensityEstimator() with every default (device 1-NN, device k-means landmarks)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench
import mellon_amd

n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 50
X = bench.gaussian_mixture(n, d, 3)
t0 = time.perf_counter()
est = mellon_amd.DensityEstimator()
dens = est.fit_predict(X)
t1 = time.perf_counter()
print(f"DensityEstimator().fit_predict({n} x {d}): {t1 - t0:.2f} s end to end "
      f"(n_landmarks={est.n_landmarks}, gp_type={est.gp_type}, evaluations={est.loss_func.n_eval})")
q = X[:50000]
t0 = time.perf_counter()
pq = est.predict(q)
print(f"predict(50000 cells): {time.perf_counter() - t0:.3f} s; |predict - fit_predict| / max = "
      f"{np.abs(pq - dens[:50000]).max() / np.abs(dens).max():.2e}")
g = est.predict.gradient(q[:2000])
print("gradient", g.shape, "finite:", bool(np.isfinite(g).all()), "log-density finite:", bool(np.isfinite(dens).all()))
