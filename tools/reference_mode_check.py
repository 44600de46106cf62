"""Distances between (a) the product's default answer (the optimum), (b) the product in reference-as-run mode,
(c) the oracle with the reference's default L-BFGS-B rule, (d) the oracle converged tightly -- C2-shaped."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mellon_amd as mellon
from oracle import mellon_oracle as mo
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
n, d, m = 20000, 20, 500
x = mo.gaussian_mixture(n, d, seed=2); nn = mo.exact_nn_distances(x)
lm = mo.compute_landmarks(x[:5000], mo.SPARSE_CHOLESKY, m, 42)
loose = mo.density_fit(x, cov_func_curry=mo.ExpQuad, landmarks=lm, nn_distances=nn)
tight = mo.density_fit(x, cov_func_curry=mo.ExpQuad, landmarks=lm, nn_distances=nn, lbfgsb_options=mo.LBFGSB_TIGHT)
est = mellon.DensityEstimator(cov_func_curry=mellon.cov.ExpQuad, landmarks=lm, nn_distances=nn); dens = est.fit_predict(x)
run = mellon.DensityEstimator(cov_func_curry=mellon.cov.ExpQuad, landmarks=lm, nn_distances=nn)
run.lbfgsb_options = "reference"; dens_run = run.fit_predict(x)
# the oracle's own early-stopped run from a start perturbed at rounding level: its self-reproducibility
pert = mo.density_fit(x, cov_func_curry=mo.ExpQuad, landmarks=lm, nn_distances=nn,
                      initial_value=loose.initial_value * (1 + 1e-13 * np.random.default_rng(0).normal(size=m)))
print(json.dumps({"product_default_vs_optimum": rel(dens, tight.log_density_x),
                  "product_default_vs_reference_default": rel(dens, loose.log_density_x),
                  "reference_mode_vs_reference_default": rel(dens_run, loose.log_density_x),
                  "reference_mode_vs_optimum": rel(dens_run, tight.log_density_x),
                  "reference_default_vs_optimum": rel(loose.log_density_x, tight.log_density_x),
                  "reference_default_vs_itself_perturbed_1e-13": rel(pert.log_density_x, loose.log_density_x),
                  "evals": {"reference_mode": run.loss_func.n_eval, "oracle_default": loose.n_eval, "oracle_perturbed": pert.n_eval,
                            "product_default": est.loss_func.n_eval}}, indent=1))
