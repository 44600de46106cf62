"""mln_kmeans at the C3 shape (1e6 x 50 -> 5000), twice; run under rocprofv3 --kernel-trace --stats for the per-kernel table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1"); os.environ.setdefault("MELLON_AMD_KM_VERBOSE", "1")
import bench
from mellon_amd import _lib
ctx = _lib.default_context()
x = bench.gaussian_mixture(1_000_000, 50, 3); xd = ctx.to_device(x)
for rep in range(2):
    t0 = time.perf_counter(); c, nit, inertia = ctx.kmeans(xd, 5000, seed=42, return_info=True)
    print(f"kmeans {rep}: {time.perf_counter() - t0:.3f} s, {nit} sweeps, inertia {inertia:.6g}", flush=True)
t0 = time.perf_counter(); nn = ctx.nn_distances(xd, xd); print(f"nn: {time.perf_counter() - t0:.3f} s")
