import sys, time
sys.path.insert(0, ".")
import numpy as np, bench
from mellon_amd import _lib
ctx = _lib.default_context()
for n in (100_000, 1_000_000):
    x = bench.gaussian_mixture(n, 50, 3); xd = ctx.to_device(x)
    ctx.kmeans(xd, 5000, seed=42)
    t0 = time.perf_counter(); c, it, inertia = ctx.kmeans(xd, 5000, seed=42, return_info=True); dt = time.perf_counter() - t0
    print(f"kmeans {n} x 50 -> 5000: {dt:.3f} s, {it} sweeps, inertia {inertia:.6g}")
