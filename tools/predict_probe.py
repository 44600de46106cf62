"""Throughput of the fused predictive mean at the C3 shape."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, bench
from mellon_amd import _lib, cov
ctx = _lib.default_context()
n, d, m = 1_000_000, 50, 5000
x = bench.gaussian_mixture(n, d, 3); c = x[:m].copy(); w = np.random.default_rng(0).normal(size=m)
xd = ctx.to_device(x)
k = cov.Matern52(25.0)
for rep in range(3):
    t0 = time.perf_counter(); out = ctx.predict_mean(k.lower(d), xd, c, w, -3.0); dt = time.perf_counter() - t0
from oracle import mellon_oracle as mo
ref = -3.0 + mo.Matern52(25.0).k(x[:2000], c) @ w
print(f"predict 1e6 x 5000 x 50: {dt*1e3:.1f} ms = {n/dt/1e6:.1f} M cells/s; err {np.abs(out[:2000]-ref).max()/np.abs(ref).max():.2e}")
