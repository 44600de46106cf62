"""Time the kernel-matrix pass alone (C3 shape) through mln_fit_prepare's stage timer."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, bench
from mellon_amd import _lib, cov
ctx = _lib.default_context()
n, d, m = 1_000_000, 50, 5000
x = bench.gaussian_mixture(n, d, 3); lm = x[:m].copy(); xd = ctx.to_device(x)
k = cov.Matern52(25.0)
for rep in range(3):
    f = ctx.fit_prepare(k.lower(d), xd, lm, 1e-6, implicit=True)
    st = f.stage_times(); f.close()
print("kernel_matrix_s", st["kernel_matrix_s"], "cholesky_s", st["cholesky_s"])
