"""Mixed-precision warm-up of mln_map_solve at C3: evaluations / time / error vs the fp32-phase tolerance."""
import os as _os; _os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")   # this tool turns experiment knobs (csrc/mln_options.h)
import os, sys, time
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
os.environ.pop("MELLON_AMD_MIXED", None)
zb, lb, nb, _, _ = fit.map_solve(z0, maxcor=50, ftol=0.0, gtol=1e-10, maxiter=2000)
fb = fit.transform(zb, est.mu)
print("best: evals", nb, "loss", lb, flush=True)
os.environ["MELLON_AMD_TRACE"] = "1"
for mixed, ftol32 in ((0, 0), (1, 1e-6), (1, 1e-7), (1, 1e-8), (1, 1e-9), (1, 1e-10)):
    os.environ["MELLON_AMD_MIXED"] = str(mixed)
    os.environ["MELLON_AMD_MIXED_FTOL"] = repr(ftol32)
    for rep in range(2):
        t0 = time.perf_counter()
        z, l, ne, ni, st = fit.map_solve(z0)
        dt = time.perf_counter() - t0
    f = fit.transform(z, est.mu)
    print(f"mixed={mixed} ftol32={ftol32:g}: evals={ne:3d} iters={ni:3d} status={st} "
          f"rel_err={np.abs(f - fb).max() / np.abs(fb).max():.2e} time={dt*1e3:.0f} ms", flush=True)
