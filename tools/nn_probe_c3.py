"""mln_nn_distances at the C3 shape (1e6 x 50 against itself), twice; run under rocprofv3 for the sweep's counters."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mellon_amd import _lib
ctx = _lib.default_context()
x = bench.gaussian_mixture(1_000_000, 50, 3); xd = ctx.to_device(x)
for rep in range(2):
    t0 = time.perf_counter(); nn = ctx.nn_distances(xd, xd); print(f"nn {rep}: {time.perf_counter() - t0:.3f} s", flush=True)
