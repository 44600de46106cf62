"""Wall-clock per pipeline stage of DensityEstimator.fit_predict at bench size."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import mellon_amd
from mellon_amd import _lib

n, d, m = int(float(sys.argv[1])), int(sys.argv[2]), int(sys.argv[3])
ctx = _lib.default_context()
x = bench.gaussian_mixture(n, d, 3)
lm = bench.make_landmarks(x, m)
xd = ctx.to_device(x)
nn = ctx.nn_distances(xd)
for rep in range(2):
    est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn)
    orig = est._prepare_attribute
    times = {}
    def timed(attr, orig=orig, times=times):
        t0 = time.perf_counter(); orig(attr); ctx.synchronize(); times[attr] = times.get(attr, 0) + time.perf_counter() - t0
    est._prepare_attribute = timed
    t0 = time.perf_counter(); est.prepare_inference(xd); t1 = time.perf_counter()
    est.run_inference(); t2 = time.perf_counter()
    est.process_inference(build_predict=False); t3 = time.perf_counter()
    print(f"rep {rep}: prepare {t1-t0:.3f} run {t2-t1:.3f} process {t3-t2:.3f} evals {est.loss_func.n_eval}")
    print("   ", {k: round(v, 4) for k, v in times.items() if v > 1e-3})
    print("   ", {k: round(v, 4) for k, v in est._fit.stage_times().items()})
    t0 = time.perf_counter(); del est; gc.collect(); ctx.synchronize(); print("    free", round(time.perf_counter() - t0, 3))
