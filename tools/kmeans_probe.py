import sys, time
sys.path.insert(0, ".")
import numpy as np, bench
from mellon_amd import _lib
ctx = _lib.default_context()
x = bench.gaussian_mixture(1_000_000, 50, 3); xd = ctx.to_device(x)
t0 = time.perf_counter(); out = ctx.kmeans(xd, 5000, seed=42); dt = time.perf_counter() - t0
print("kmeans 1e6x50 m=5000:", round(dt, 2), "s", [type(o) for o in out] if isinstance(out, tuple) else type(out), out[1:] if isinstance(out, tuple) else "")
