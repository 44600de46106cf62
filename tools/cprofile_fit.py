"""cProfile of one C3 fit_predict from DEVICE cells and from HOST cells (pageable / page-locked): where the host-to-host
step's extra milliseconds go.   python tools/cprofile_fit.py [device|host|pinned]"""
import os, sys, cProfile, pstats, gc, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELLON_AMD_MIXED", "0")
import bench, mellon_amd
from mellon_amd import _lib
ctx = _lib.default_context()
n, d, m = 1_000_000, 50, 5000
x = bench.gaussian_mixture(n, d, 3); lm, _ = bench.make_landmarks(x, m, "device", ctx); xd = ctx.to_device(x); nn = ctx.nn_distances(xd)
mode = sys.argv[1] if len(sys.argv) > 1 else "device"
def run(xin):
    est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
    out = est.fit_predict(xin)
    global last_stages
    last_stages = est._fit.stage_times()
    est._fit.close()
    return out
def go(xin):
    run(xin); run(xin); gc.collect()
    t0 = time.perf_counter(); run(xin); print(mode, "step ms", 1e3 * (time.perf_counter() - t0))
    pr = cProfile.Profile(); pr.enable(); run(xin); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
    print({k: round(1e3 * v, 2) for k, v in last_stages.items() if k.endswith("_s")})
if mode == "device": go(xd)
elif mode == "host": go(x)
else:
    with ctx.pinned(x): go(x)
