"""Time the per-output-sigma landmark solve (mln_sparse_solve_noise) at config-5 size against the scalar solve."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import mellon_amd
from mellon_amd import _lib
from mellon_amd.conditional import _sparse_solve_per_output

n, d, m, p = 200_000, 50, 2000, int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(5)
X = rng.normal(size=(n, d))
Y = np.sin(X @ rng.normal(size=(d, p)) / np.sqrt(d)) + 0.1 * rng.normal(size=(n, p))
xu = X[rng.choice(n, m, replace=False)]
from mellon_amd.cov import Matern52
cov = Matern52(ls=8.0)
ctx = _lib.default_context()
desc = cov.lower(d)
xd = ctx.to_device(X) if hasattr(ctx, "to_device") else X
for label, levels in (("1 level", 1), ("16 levels", 16), ("p levels", p)):
    sigma = 0.1 * (1 + (np.arange(p) % levels) / levels)
    for rep in range(2):
        t0 = time.perf_counter()
        W = _sparse_solve_per_output(ctx, desc, xd, xu, Y, 0.0, sigma, 1e-6)
        dt = time.perf_counter() - t0
    print(f"per-output sigma, {label}: {dt:.3f} s")
t0 = time.perf_counter()
W0 = ctx.sparse_solve(desc, xd, xu, Y, 0.0, 0.1, 1e-6)
print(f"scalar sigma: {time.perf_counter() - t0:.3f} s")
sigma = np.full(p, 0.1)
W1 = _sparse_solve_per_output(ctx, desc, xd, xu, Y, 0.0, sigma, 1e-6)
print("max |W_per_output(const) - W_scalar| / max|W| =", np.abs(W1 - W0).max() / np.abs(W0).max())

# whole per-gene fit with smoothed observation variance: weights, leverage (n x p), HC3 residuals, variance weights
sigma = 0.1 * (1 + np.arange(p) / p)
for rep in range(2):
    t0 = time.perf_counter()
    est = mellon_amd.FunctionEstimator(sigma=sigma, n_landmarks=m, landmarks=xu, ls=8.0, obs_variance=True).fit(X, Y)
    print(f"FunctionEstimator(sigma[p], obs_variance=True).fit: {time.perf_counter() - t0:.3f} s")
t0 = time.perf_counter()
est = mellon_amd.FunctionEstimator(sigma=0.1, n_landmarks=m, landmarks=xu, ls=8.0, obs_variance=True).fit(X, Y)
print(f"FunctionEstimator(scalar sigma, obs_variance=True).fit: {time.perf_counter() - t0:.3f} s")
