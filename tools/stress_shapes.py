"""Odd shapes / kernels through the full estimator with the mixed-precision path forced on, vs the oracle."""
import os as _os; _os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")   # this tool turns experiment knobs (csrc/mln_options.h)
import os, sys
os.environ["MELLON_AMD_MIXED_MIN_ELEMS"] = "0"
sys.path.insert(0, ".")
import numpy as np
import mellon_amd
from oracle import mellon_oracle as mo
rng = np.random.default_rng(123)
worst = 0.0
for trial in range(14):
    n = int(rng.integers(300, 5000)); d = int(rng.integers(1, 12)); m = int(rng.integers(5, min(400, n // 3)))
    kname = ["Matern52", "Matern32", "ExpQuad", "Exponential", "RatQuad"][trial % 5]
    x = mo.gaussian_mixture(n, d, seed=100 + trial)
    lm = x[rng.choice(n, m, replace=False)]
    nn = mo.exact_nn_distances(x)
    kw = dict(landmarks=lm, nn_distances=nn)
    ref = mo.density_fit(x, cov_func_curry=getattr(mo, kname), lbfgsb_options=mo.LBFGSB_TIGHT, **kw)
    est = mellon_amd.DensityEstimator(cov_func_curry=getattr(mellon_amd.cov, kname), **kw)
    dens = est.fit_predict(x)
    st = est._fit.stage_times()
    err = np.abs(dens - ref.log_density_x).max() / np.abs(ref.log_density_x).max()
    perr = np.abs(est.predict(x[:200]) - dens[:200]).max() / np.abs(dens).max()
    g = est.predict.gradient(x[:50]); gr = ref.predict.gradient(x[:50])
    gerr = np.abs(g - gr).max() / max(np.abs(gr).max(), 1e-12)
    worst = max(worst, err)
    print(f"n={n} d={d} m={m} {kname}: rel_err={err:.2e} predict={perr:.1e} grad={gerr:.1e} evals={est.loss_func.n_eval} fp32={int(st['objective32_launches'])}", flush=True)
print("worst", worst)
assert worst < 1e-5
