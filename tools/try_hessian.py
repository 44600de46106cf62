"""Time mln_predict_hessian at a realistic size."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mellon_amd import _lib
from mellon_amd.cov import Matern52

ctx = _lib.default_context()
rng = np.random.default_rng(0)
for n, m, d, label, cov in ((20_000, 5000, 50, "Matern52", Matern52(ls=8.0)),
                            (20_000, 5000, 51, "time-sensitive product",
                             Matern52(ls=8.0, active_dims=slice(0, 50)) * Matern52(ls=1.5, active_dims=50))):
    X, C, w = rng.normal(size=(n, d)), rng.normal(size=(m, d)), rng.normal(size=m)
    desc = cov.lower(d)
    for rep in range(2):
        t0 = time.perf_counter()
        H = ctx.predict_hessian(desc, X, C, w)
        dt = time.perf_counter() - t0
    flops = 2.0 * n * m * d * d
    print(f"{label}: {n} x {m} x {d}: {dt:.3f} s ({n / dt:,.0f} cells/s, pair GEMM {flops / dt / 1e12:.1f} TFLOP/s incl. transfers)")
