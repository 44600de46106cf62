"""Diagnose the cases tools/robustness_sweep_large.py flags, at a size the oracle finishes: every solve mode against the
oracle's optimum.   python tools/robustness_diag.py [n] [m]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")
import numpy as np
import mellon_amd
from mellon_amd import _lib
from oracle import mellon_oracle as mo
import importlib.util
spec = importlib.util.spec_from_file_location("rsl", os.path.join(os.path.dirname(os.path.abspath(__file__)), "robustness_sweep_large.py"))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 500
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
cases = {"tree d=20": trajectories(n, 20), "tree d=10": trajectories(n, 10), "t3 d=20": rng.standard_t(3, size=(n, 20))}
for name, x in cases.items():
    x = np.ascontiguousarray(x)
    nn = ctx.nn_distances(x, x)
    lm = ctx.kmeans(x[:min(n, 100000)], m, seed=42)
    t0 = time.perf_counter()
    ref = mo.density_fit(x, landmarks=lm, nn_distances=nn, lbfgsb_options=mo.LBFGSB_TIGHT)
    print(f"== {name}: oracle {time.perf_counter() - t0:.1f} s, {ref.n_eval} evals, loss {ref.loss:.10g}, log-density range [{ref.log_density_x.min():.1f}, {ref.log_density_x.max():.1f}], "
          f"nn range [{nn.min():.2e}, {nn.max():.2e}], ls {ref.ls:.3g}", flush=True)
    scale = np.abs(ref.log_density_x).max()
    modes = [("default fp64", {"MELLON_AMD_MIXED": "0"}),
             ("no subsample", {"MELLON_AMD_MIXED": "0", "MELLON_AMD_SUBSAMPLE": "0"}),
             ("no rebuild", {"MELLON_AMD_MIXED": "0", "MELLON_AMD_REBUILD": "0"}),
             ("plain fp64", {"MELLON_AMD_MIXED": "0", "MELLON_AMD_SUBSAMPLE": "0", "MELLON_AMD_REBUILD": "0"}),
             ("plain fp64, fp64 Gram", {"MELLON_AMD_MIXED": "0", "MELLON_AMD_SUBSAMPLE": "0", "MELLON_AMD_REBUILD": "0", "MELLON_AMD_GRAM_I8": "0"}),
             ("mixed", {"MELLON_AMD_MIXED_MIN_ELEMS": "0"}),
             ("reference-as-run (SciPy over the device objective)", None)]
    for mode, env in modes:
        keys = ("MELLON_AMD_MIXED", "MELLON_AMD_SUBSAMPLE", "MELLON_AMD_REBUILD", "MELLON_AMD_GRAM_I8", "MELLON_AMD_MIXED_MIN_ELEMS")
        saved = {k: os.environ.get(k) for k in keys}
        for k in keys: os.environ.pop(k, None)
        try:
            est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
            if env is None:
                os.environ["MELLON_AMD_MIXED"] = "0"
                est.lbfgsb_options = "reference"
            else:
                os.environ.update(env)
            t0 = time.perf_counter()
            dens = est.fit_predict(x)
            dt = time.perf_counter() - t0
            st = est._fit.stage_times()
            loss = float(est.losses[-1]) if getattr(est, "losses", None) else float("nan")
            err = np.abs(dens - ref.log_density_x).max() / scale
            print(f"   {mode:52s} evals {est.loss_func.n_eval:5d} passes {st.get('objective_pass_equivalents', float('nan')):6.1f} loss {loss:.10g} "
                  f"rel_max vs oracle {err:.2e}  status {getattr(est.opt_state, 'status', getattr(est.opt_state, 'success', '?'))}  {1e3 * dt:.0f} ms", flush=True)
            est._fit.close()
        except Exception as e:      # noqa: BLE001
            print(f"   {mode:52s} FAILED {type(e).__name__}: {str(e)[:120]}", flush=True)
        finally:
            for k, v in saved.items():
                os.environ.pop(k, None)
                if v is not None: os.environ[k] = v
