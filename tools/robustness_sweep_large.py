"""The solver's path shortcuts (subsample phase, importance-sampled second preconditioner, step rules) were tuned on
Gaussian mixtures.  This sweep runs the default fit at FULL scale (1e6 cells) on data of other shapes and checks
size-independent properties (no oracle finishes at this size):
  * default path == the plain path (MELLON_AMD_SUBSAMPLE=0, MELLON_AMD_REBUILD=0) of the same fp64 solve: the shortcuts
    change the iteration path, not the optimum;
  * the opt-in mixed-precision solve (MELLON_AMD_MIXED=1) == the fp64 solve (the product default since round 5);
  * predict(X) == fit_predict(X);
and reports pass counts per case.      python tools/robustness_sweep_large.py [n]      (on a GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")
import numpy as np
import mellon_amd
from mellon_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(11)
ctx = _lib.default_context()


def trajectories(n, d, branches=6):
    """Diffusion-map-like coordinates: cells along a branching tree of smooth curves in a 3-D latent space, embedded by a
    random smooth map, column k scaled by 0.8^k (the geometric eigenvalue decay of a diffusion map), unevenly populated."""
    t = rng.beta(0.7, 1.3, size=n)                         # pseudo-time, dense near the root
    b = rng.integers(0, branches, size=n)
    dirs = rng.normal(size=(branches, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    bend = rng.normal(size=(branches, 3)) * 0.5
    z = t[:, None] * dirs[b] + (t ** 2)[:, None] * bend[b] + 0.02 * (1 + 3 * t)[:, None] * rng.normal(size=(n, 3))
    W1 = rng.normal(size=(3, d)); W2 = rng.normal(size=(3, d))
    x = np.tanh(z @ W1) + 0.3 * np.sin(2.0 * z @ W2)
    return np.ascontiguousarray(x * (0.8 ** np.arange(d))[None, :])


def mixture(n, d, k=10):
    means = rng.normal(0, 3, size=(k, d)); sig = rng.uniform(0.5, 1.5, size=k)
    c = rng.integers(0, k, size=n)
    return means[c] + rng.normal(size=(n, d)) * sig[c][:, None]


cases = {
    "diffusion-map-like tree, d = 20": lambda: trajectories(n, 20),
    "diffusion-map-like tree, d = 10": lambda: trajectories(n, 10),
    "mixture d = 20, 10 % duplicated cells": lambda: (lambda x: (x.__setitem__(slice(0, n // 10), x[n // 2:n // 2 + n // 10]), x)[1])(mixture(n, 20)),
    "heavy tails (t3), d = 20": lambda: rng.standard_t(3, size=(n, 20)),
    "two scales: tight cluster of 5 % + broad cloud, d = 20": lambda: np.concatenate([0.01 * rng.normal(size=(n // 20, 20)) + 4.0, mixture(n - n // 20, 20)]),
}
m = 2000
worst = 0.0
for name, make in cases.items():
    x = np.ascontiguousarray(make(), dtype=np.float64)
    xd = ctx.to_device(x)
    nn = ctx.nn_distances(xd, xd)
    lm = ctx.kmeans(x[:100_000], m, seed=42)
    out = {}
    for mode, env in (("default fp64", {"MELLON_AMD_MIXED": "0"}),
                      ("plain fp64", {"MELLON_AMD_MIXED": "0", "MELLON_AMD_SUBSAMPLE": "0", "MELLON_AMD_REBUILD": "0"}),
                      ("mixed", {"MELLON_AMD_MIXED": "1"})):
        saved = {k: os.environ.get(k) for k in ("MELLON_AMD_MIXED", "MELLON_AMD_SUBSAMPLE", "MELLON_AMD_REBUILD")}
        for k in saved: os.environ.pop(k, None)
        os.environ.update(env)
        try:
            t0 = time.perf_counter()
            est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
            dens = est.fit_predict(xd)
            dt = time.perf_counter() - t0
            st = est._fit.stage_times()
            out[mode] = (dens, est.loss_func.n_eval, st.get("objective_pass_equivalents", float("nan")), dt, est,
                         (int(getattr(est.opt_state, "status", -1)), int(st.get("precond_rebuilds", 0)), int(st.get("precond_rebuilds_declined", 0)),
                          int(st.get("precond_reverts", 0)), int(st.get("start_halvings", 0))))
        finally:
            for k, v in saved.items():
                os.environ.pop(k, None)
                if v is not None: os.environ[k] = v
    ref = out["plain fp64"][0]
    scale = np.abs(ref).max()
    e_path = np.abs(out["default fp64"][0] - ref).max() / scale
    e_mixed = np.abs(out["mixed"][0] - out["default fp64"][0]).max() / scale
    est = out["default fp64"][4]
    e_pred = np.abs(est.predict(x[:20000]) - out["default fp64"][0][:20000]).max() / scale
    worst = max(worst, e_path, e_mixed, e_pred)
    print(f"{name:58s} default {out['default fp64'][1]:3d} evals / {out['default fp64'][2]:5.1f} passes / {1e3 * out['default fp64'][3]:6.1f} ms | "
          f"plain {out['plain fp64'][1]:3d} evals / {out['plain fp64'][2]:5.1f} passes | default vs plain {e_path:.1e}  mixed vs fp64 {e_mixed:.1e}  "
          f"predict vs fit {e_pred:.1e}  finite {bool(np.isfinite(ref).all())}  (status, rebuilds, declined, reverts, start halvings) default {out['default fp64'][5]} "
          f"plain {out['plain fp64'][5]} mixed {out['mixed'][5]}", flush=True)
    for v in out.values(): v[4]._fit.close()
    xd.free()
print("worst", worst)
