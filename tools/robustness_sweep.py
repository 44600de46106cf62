"""Default-argument fits on awkward data sets (duplicates, units, offsets, near-degenerate dimensions, tiny and skewed
clusters) against the oracle with the same landmarks / nn distances: the drop-in path end to end."""
import sys
sys.path.insert(0, ".")
import numpy as np
import mellon_amd
from oracle import mellon_oracle as mo

rng = np.random.default_rng(7)
cases = {}
base = mo.gaussian_mixture(12000, 8, seed=3)
cases["plain"] = base
d = base.copy(); d[:3000] = d[3000:6000]; cases["25% duplicated cells"] = d
cases["units of 1e4, offset 1e6"] = base * 1e4 + 1e6
cases["units of 1e-5"] = base * 1e-5
s = base.copy(); s[:, 5:] *= 1e-6; cases["three near-degenerate columns"] = s
t = np.concatenate([base[:11000], 50 + 1e-3 * rng.normal(size=(1000, 8))]); cases["tight far cluster"] = t
cases["uniform cube"] = rng.uniform(-1, 1, size=(12000, 8))
cases["heavy tails (t3)"] = rng.standard_t(3, size=(12000, 8))
cases["1-D"] = np.sort(rng.normal(size=(12000, 1)), axis=0)
worst = 0.0
for name, x in cases.items():
    x = np.ascontiguousarray(x, dtype=np.float64)
    try:
        est = mellon_amd.DensityEstimator(n_landmarks=400)
        dens = est.fit_predict(x)
        nn = np.asarray(est.nn_distances)
        ref = mo.density_fit(x, landmarks=np.asarray(est.landmarks), nn_distances=nn, lbfgsb_options=mo.LBFGSB_TIGHT)
        nn_ref = mo.validate_nn_distances(mo.exact_nn_distances(x))
        e_nn = np.abs(nn / nn_ref - 1).max()
        err = np.abs(dens - ref.log_density_x).max() / np.abs(ref.log_density_x).max()
        worst = max(worst, err)
        print(f"{name:32s} rel_err {err:.2e}  nn vs tree {e_nn:.1e}  evals {est.loss_func.n_eval}  finite {bool(np.isfinite(dens).all())}", flush=True)
    except Exception as e:      # noqa: BLE001
        print(f"{name:32s} FAILED {type(e).__name__}: {e}", flush=True)
        worst = float("inf")
print("worst", worst)
