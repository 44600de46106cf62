"""Pass count vs accuracy of mln_map_solve at C3 for a few (maxcor, ftol, gtol) settings."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, mellon_amd
from mellon_amd import _lib
ctx = _lib.default_context()
n, d, m = 1_000_000, 50, 5000
x = bench.gaussian_mixture(n, d, 3); lm = bench.make_landmarks(x, m); xd = ctx.to_device(x); nn = ctx.nn_distances(xd)
est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn)
est.prepare_inference(xd)
fit = est._fit
z0 = est.initial_value
zb, lb, nb, _, _ = fit.map_solve(z0, maxcor=50, ftol=0.0, gtol=1e-10, maxiter=2000)
fb = fit.transform(zb, est.mu)
print("best: evals", nb, "loss", lb)
for maxcor in (10, 30, 60):
    for ftol, gtol in ((1e-9, 1e-5), (1e-10, 1e-5), (1e-11, 1e-6), (1e-12, 1e-6), (1e-13, 1e-7)):
        t0 = time.perf_counter()
        z, l, ne, ni, st = fit.map_solve(z0, maxcor=maxcor, ftol=ftol, gtol=gtol)
        dt = time.perf_counter() - t0
        f = fit.transform(z, est.mu)
        print(f"maxcor={maxcor:2d} ftol={ftol:g} gtol={gtol:g}: evals={ne:3d} iters={ni:3d} status={st} "
              f"rel_err={np.abs(f - fb).max() / np.abs(fb).max():.2e} time={dt*1e3:.0f} ms")
