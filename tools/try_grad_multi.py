"""Composite-kernel predictor gradient: GEMM route against the general kernel and the oracle; timing."""
import os as _os; _os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")   # this tool turns experiment knobs (csrc/mln_options.h)
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mellon_amd import _lib
from mellon_amd.cov import ExpQuad, Matern32, Matern52, RatQuad

ctx = _lib.default_context()
rng = np.random.default_rng(0)


def check(cov, ocov, n, m, d, label):
    X, C, w = rng.normal(size=(n, d)), rng.normal(size=(m, d)), rng.normal(size=m)
    desc = cov.lower(d)
    g1 = ctx.predict_gradient(desc, X, C, w)
    os.environ["MELLON_AMD_GRAD_NO_GEMM"] = "1"
    g0 = ctx.predict_gradient(desc, X, C, w)
    del os.environ["MELLON_AMD_GRAD_NO_GEMM"]
    err = np.abs(g1 - g0).max() / np.abs(g0).max()
    print(f"{label}: gemm vs general rel max {err:.2e}")
    return err


d = 12
t = Matern52(ls=2.0, active_dims=slice(0, d - 1)) * Matern52(ls=1.5, active_dims=d - 1)
check(t, None, 700, 300, d, "time-sensitive product")
check(Matern32(ls=1.3) + 0.5 * ExpQuad(ls=2.0, active_dims=[0, 3, 5]), None, 513, 259, d, "sum with scalar product")
check((Matern52(ls=2.0) * RatQuad(alpha=1.5, ls=3.0)) ** 2.0, None, 300, 300, d, "power of product")
check(Matern52(ls=2.0, active_dims=slice(0, 6)) * ExpQuad(ls=1.0, active_dims=slice(4, 9)) + Matern32(ls=0.7), None,
      400, 400, d, "three leaves, overlapping dims")

n, m, d = 200_000, 5000, 51
X, C, w = rng.normal(size=(n, d)), rng.normal(size=(m, d)), rng.normal(size=m)
t = Matern52(ls=8.0, active_dims=slice(0, d - 1)) * Matern52(ls=1.5, active_dims=d - 1)
desc = t.lower(d)
xd, cd = ctx.to_device(X), C
for label, env in (("gemm", None), ("general", "1")):
    if env:
        os.environ["MELLON_AMD_GRAD_NO_GEMM"] = env
    for rep in range(2):
        t0 = time.perf_counter()
        g = ctx.predict_gradient(desc, xd, cd, w)
        dt = time.perf_counter() - t0
    print(f"2e5 x 5000 x 51 product kernel, {label}: {dt:.3f} s")
    os.environ.pop("MELLON_AMD_GRAD_NO_GEMM", None)
