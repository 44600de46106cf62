"""Steps before the hot path at C3 shape, each timed on its own: device k-means landmarks (sweeps, inertia), exact 1-NN
distances, the rank diagnostic.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel table.

    python tools/prepath_probe.py [n] [m] [what]      what: any of k (k-means), n (1-NN), r (rank), default all
"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np

import bench
from mellon_amd import _lib, cov

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
what = sys.argv[3] if len(sys.argv) > 3 else "knr"
d = 50
ctx = _lib.default_context()
x = bench.gaussian_mixture(n, d, 3)
xd = ctx.to_device(x)
reps = 2
if "k" in what:
    for rep in range(reps):
        t0 = time.perf_counter()
        c, nit, inertia = ctx.kmeans(xd, m, seed=42, return_info=True)
        print(f"kmeans {n}x{d} -> {m}: {time.perf_counter() - t0:.3f} s, sweeps {nit}, inertia {inertia:.6e}", flush=True)
if "n" in what:
    for rep in range(reps):
        t0 = time.perf_counter()
        nn = ctx.nn_distances(xd)
        print(f"nn_distances {n}x{d}: {time.perf_counter() - t0:.3f} s, mean {nn.mean():.6f} min {nn.min():.3e}", flush=True)
if "r" in what:
    lm = x[:: max(1, n // m)][:m].copy()
    k = cov.Matern52(25.0)
    f = ctx.fit_prepare(k.lower(d), xd, lm, 1e-6, implicit=True)
    for rep in range(reps):
        t0 = time.perf_counter()
        r, smax = f.gram_rank(0.5)
        print(f"gram_rank m={m}: {time.perf_counter() - t0:.3f} s, rank {r}, sigma_max {smax:.6e}", flush=True)
    f.close()
