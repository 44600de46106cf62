"""Throughput of the fused predictor gradient (mln_predict_gradient)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mellon_amd import _lib, cov

ctx = _lib.default_context()
rng = np.random.default_rng(0)
for n, m, d in [(100_000, 1000, 20), (200_000, 5000, 50)]:
    x, c, w = rng.normal(size=(n, d)), rng.normal(size=(m, d)), rng.normal(size=m)
    k = cov.Matern52(1.5 * np.sqrt(d))
    xd = ctx.to_device(x) if hasattr(ctx, "to_device") else x
    ctx.predict_gradient(k.lower(d), xd, c, w)
    t0 = time.perf_counter(); g = ctx.predict_gradient(k.lower(d), xd, c, w); t1 = time.perf_counter()
    t2 = time.perf_counter(); mu = ctx.predict_mean(k.lower(d), xd, c, w, 0.0); t3 = time.perf_counter()
    print(f"n={n} m={m} d={d}: gradient {t1 - t0:.3f}s ({n / (t1 - t0):.3g} cells/s, "
          f"{n * m * d * 6 / (t1 - t0) / 1e12:.2f} Tflop/s-equiv), mean {t3 - t2:.3f}s", flush=True)
