"""Full-scale (1e6 x d, m = 2000) diagnosis of the cases tools/robustness_sweep_large.py flags: every solve mode, judged by
what needs no oracle -- the loss and the gradient of the (strictly convex) objective at the returned point, evaluated by
the plain fp64 pass (fit.objective).   python tools/robustness_diag_large.py [case ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")
import numpy as np
import mellon_amd
from mellon_amd import _lib

n, m = int(os.environ.get("N", 1_000_000)), int(os.environ.get("M", 2000))
rng = np.random.default_rng(11)


def trajectories(n, d, branches=6):
    t = rng.beta(0.7, 1.3, size=n)
    b = rng.integers(0, branches, size=n)
    dirs = rng.normal(size=(branches, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    bend = rng.normal(size=(branches, 3)) * 0.5
    z = t[:, None] * dirs[b] + (t ** 2)[:, None] * bend[b] + 0.02 * (1 + 3 * t)[:, None] * rng.normal(size=(n, 3))
    W1 = rng.normal(size=(3, d)); W2 = rng.normal(size=(3, d))
    x = np.tanh(z @ W1) + 0.3 * np.sin(2.0 * z @ W2)
    return np.ascontiguousarray(x * (0.8 ** np.arange(d))[None, :])


ctx = _lib.default_context()
makers = {"tree20": lambda: trajectories(n, 20), "tree10": lambda: trajectories(n, 10), "t3": lambda: rng.standard_t(3, size=(n, 20))}
KEYS = ("MELLON_AMD_MIXED", "MELLON_AMD_SUBSAMPLE", "MELLON_AMD_REBUILD", "MELLON_AMD_GRAM_I8", "MELLON_AMD_EXPLICIT_LINV", "MELLON_AMD_TRACE")
for name in (sys.argv[1:] or list(makers)):
    x = np.ascontiguousarray(makers[name]())
    xd = ctx.to_device(x)
    nn = ctx.nn_distances(xd, xd)
    lm = ctx.kmeans(x[:100000], m, seed=42)
    print(f"== {name}: nn range [{nn.min():.2e}, {nn.max():.2e}]", flush=True)
    modes = [("default fp64", {"MELLON_AMD_MIXED": "0"}),
             ("no subsample", {"MELLON_AMD_MIXED": "0", "MELLON_AMD_SUBSAMPLE": "0"}),
             ("no rebuild", {"MELLON_AMD_MIXED": "0", "MELLON_AMD_REBUILD": "0"}),
             ("plain fp64", {"MELLON_AMD_MIXED": "0", "MELLON_AMD_SUBSAMPLE": "0", "MELLON_AMD_REBUILD": "0"}),
             ("plain, fp64 Gram", {"MELLON_AMD_MIXED": "0", "MELLON_AMD_SUBSAMPLE": "0", "MELLON_AMD_REBUILD": "0", "MELLON_AMD_GRAM_I8": "0"}),
             ("plain, fp64 Gram, solves instead of Lp^-1", {"MELLON_AMD_MIXED": "0", "MELLON_AMD_SUBSAMPLE": "0", "MELLON_AMD_REBUILD": "0", "MELLON_AMD_GRAM_I8": "0", "MELLON_AMD_EXPLICIT_LINV": "0"}),
             ("default, solves instead of Lp^-1", {"MELLON_AMD_MIXED": "0", "MELLON_AMD_EXPLICIT_LINV": "0"}),
             ("mixed (product default)", {}),
             ("reference-as-run (SciPy)", None)]
    base = None
    for mode, env in modes:
        saved = {k: os.environ.get(k) for k in KEYS}
        for k in KEYS: os.environ.pop(k, None)
        try:
            est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
            if env is None:
                os.environ["MELLON_AMD_MIXED"] = "0"; est.lbfgsb_options = "reference"
            else:
                os.environ.update(env)
            t0 = time.perf_counter()
            dens = est.fit_predict(xd)
            dt = time.perf_counter() - t0
            st = est._fit.stage_times()
            z = np.asarray(est.pre_transformation)
            loss, grad = est._fit.objective(z)
            if base is None and "plain, fp64 Gram, solves" in mode: base = dens
            print(f"   {mode:44s} evals {est.loss_func.n_eval:5d} passes {st.get('objective_pass_equivalents', float('nan')):6.1f} status {getattr(est.opt_state, 'status', '?')} "
                  f"loss(z) {loss:.12g} |grad_z|max {np.abs(grad).max():.3e} |z|max {np.abs(z).max():.3g} dens [{dens.min():.1f}, {dens.max():.1f}] {1e3 * dt:.0f} ms", flush=True)
            est._fit.close()
        except Exception as e:      # noqa: BLE001
            print(f"   {mode:44s} FAILED {type(e).__name__}: {str(e)[:140]}", flush=True)
        finally:
            for k, v in saved.items():
                os.environ.pop(k, None)
                if v is not None: os.environ[k] = v
    xd.free()
