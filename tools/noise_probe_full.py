"""Time the full-GP per-output-sigma fit (mln_full_conditional_noise) against the per-level route."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import mellon_amd
from mellon_amd import conditional

n, d, p = 3000, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.default_rng(5)
X = rng.normal(size=(n, d))
Y = np.sin(X @ rng.normal(size=(d, p)) / np.sqrt(d)) + 0.1 * rng.normal(size=(n, p))
sigma = 0.1 * (1 + np.arange(p) / p)
for label, thresh in (("spectral", 8), ("per level", 10 ** 9)):
    conditional.SPECTRAL_MIN_LEVELS = thresh
    for rep in range(2):
        t0 = time.perf_counter()
        est = mellon_amd.FunctionEstimator(sigma=sigma, n_landmarks=0, ls=5.0, obs_variance=True).fit(X, Y)
        dt = time.perf_counter() - t0
    print(f"full GP n={n}, {p} noise levels, obs_variance=True, {label}: {dt:.3f} s")
    out = (est.predict(X[:200]), est.get_obs_variance(X[:200]))
    if label == "spectral":
        keep = out
    else:
        print("max rel diff predict / obs_variance:", np.abs(out[0] - keep[0]).max() / np.abs(keep[0]).max(),
              np.abs(out[1] - keep[1]).max() / np.abs(keep[1]).max())
