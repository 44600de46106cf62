"""The two workloads of tools/robustness_sweep_large.py on which the MAP solve is slow -- the diffusion-map-like tree
(d = 20) and heavy tails (t3, d = 20) -- one fit per environment given on the command line, against the plain solve
(no subsample phase, no rebuild) of the same fp64 objective.

    python tools/hard_cases.py [n] [case,...] -- "" "MELLON_AMD_REBUILD_RANGE=1e300" "A=1 B=2" ...

Prints per (case, environment): evaluations, pass-equivalents, wall ms, rebuilds / declined / reverts, relative distance
of the log-density from the plain solve's, and the gradient norm at the returned point."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")
import numpy as np
import mellon_amd
from mellon_amd import _lib

argv = sys.argv[1:]
envs = [""]
if "--" in argv:
    k = argv.index("--")
    envs = argv[k + 1:] or [""]
    argv = argv[:k]
n = int(float(argv[0])) if argv else 1_000_000
which = argv[1].split(",") if len(argv) > 1 else ["tree20", "t3"]
m = int(os.environ.get("HARD_M", "2000"))
rng = np.random.default_rng(11)
ctx = _lib.default_context()


def trajectories(n, d, branches=6):
    t = rng.beta(0.7, 1.3, size=n)
    b = rng.integers(0, branches, size=n)
    dirs = rng.normal(size=(branches, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    bend = rng.normal(size=(branches, 3)) * 0.5
    z = t[:, None] * dirs[b] + (t ** 2)[:, None] * bend[b] + 0.02 * (1 + 3 * t)[:, None] * rng.normal(size=(n, 3))
    W1 = rng.normal(size=(3, d)); W2 = rng.normal(size=(3, d))
    x = np.tanh(z @ W1) + 0.3 * np.sin(2.0 * z @ W2)
    return np.ascontiguousarray(x * (0.8 ** np.arange(d))[None, :])


cases = {"tree20": lambda: trajectories(n, 20), "tree10": lambda: trajectories(n, 10),
         "t3": lambda: rng.standard_t(3, size=(n, 20))}
KEYS = ("MELLON_AMD_MIXED", "MELLON_AMD_SUBSAMPLE", "MELLON_AMD_REBUILD")


def run(xd, lm, nn, env):
    saved = dict(os.environ)
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ["MELLON_AMD_MIXED"] = "0"
    for kv in env.split():
        k, v = kv.split("=", 1)
        os.environ[k] = v
    try:
        t0 = time.perf_counter()
        est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
        dens = est.fit_predict(xd)
        dt = time.perf_counter() - t0
        st = est._fit.stage_times()
        z = np.asarray(est.pre_transformation)
        loss, grad = est._fit.objective(z)
        info = (est.loss_func.n_eval, st.get("objective_pass_equivalents", float("nan")), 1e3 * dt, int(st.get("precond_rebuilds", 0)),
                int(st.get("precond_rebuilds_declined", 0)), int(st.get("precond_reverts", 0)), int(getattr(est.opt_state, "status", -1)),
                loss, float(np.abs(grad).max()))
        est._fit.close()
        return dens, info
    finally:
        os.environ.clear()
        os.environ.update(saved)


for name in which:
    x = np.ascontiguousarray(cases[name](), dtype=np.float64)
    xd = ctx.to_device(x)
    nn = ctx.nn_distances(xd, xd)
    lm = ctx.kmeans(x[:100_000], m, seed=42)
    ref, info = run(xd, lm, nn, "MELLON_AMD_SUBSAMPLE=0 MELLON_AMD_REBUILD=0")
    scale = np.abs(ref).max()
    print(f"{name:7s} plain: {info[0]} evals / {info[1]:.1f} passes / {info[2]:.0f} ms  status {info[6]} loss {info[7]:.10g} |g|max {info[8]:.2e}", flush=True)
    for env in envs:
        dens, info = run(xd, lm, nn, env)
        print(f"{name:7s} [{env or 'default'}]: {info[0]} evals / {info[1]:.1f} passes / {info[2]:.0f} ms  rebuilds {info[3]} declined {info[4]} reverts {info[5]} "
              f"status {info[6]} loss {info[7]:.10g} |g|max {info[8]:.2e}  vs plain {np.abs(dens - ref).max() / scale:.2e}", flush=True)
    xd.free()
