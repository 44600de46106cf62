"""Which estimator paths accept a Covariance subclass with a Python k()?  (run on the GPU box)"""
import traceback
import numpy as np
import mellon_amd as mellon
from oracle import mellon_oracle as mo


class Cauchy(mellon.Covariance):
    def __init__(self, ls=1.0, active_dims=None):
        super().__init__()
        self.ls = ls
        self.active_dims = active_dims

    def k(self, x, y):
        d2 = ((x[:, None, :] - y[None, :, :]) ** 2).sum(-1)
        return 1.0 / (1.0 + d2 / self.ls ** 2)


def attempt(name, fn):
    try:
        out = fn()
        print(f"OK   {name}: {out}")
    except Exception as e:
        print(f"FAIL {name}: {type(e).__name__}: {e}")
        traceback.print_exc(limit=3)


rng = np.random.default_rng(0)
n, d, m = 3000, 5, 200
x = mo.gaussian_mixture(n, d, seed=1)
nn = mo.exact_nn_distances(x)
ls = mo.compute_ls(nn)
cov = Cauchy(ls)

def de():
    est = mellon.DensityEstimator(cov_func=cov, n_landmarks=m)
    f = est.fit_predict(x)
    p = est.predict(x[:100])
    return float(np.abs(p - f[:100]).max())
attempt("DensityEstimator fit_predict/predict", de)

def de_unc():
    est = mellon.DensityEstimator(cov_func=cov, n_landmarks=m, predictor_with_uncertainty=True)
    est.fit(x)
    return float(est.predict.uncertainty(x[:10]).mean())
attempt("DensityEstimator uncertainty", de_unc)

def de_full():
    est = mellon.DensityEstimator(cov_func=cov, n_landmarks=0, gp_type="full")
    return float(est.fit_predict(x[:800]).mean())
attempt("DensityEstimator full GP", de_full)

def de_nys():
    est = mellon.DensityEstimator(cov_func=cov, n_landmarks=m, gp_type="sparse_nystroem", rank=50)
    return float(est.fit_predict(x).mean())
attempt("DensityEstimator sparse_nystroem", de_nys)

def fe():
    y = np.sin(x[:, :3]) + 0.1 * rng.normal(size=(n, 3))
    est = mellon.FunctionEstimator(cov_func=cov, n_landmarks=m, sigma=0.1)
    return float(np.abs(est.fit_predict(x, y, x) - y).mean())
attempt("FunctionEstimator", fe)

def fe_unc():
    y = np.sin(x[:, :3]) + 0.1 * rng.normal(size=(n, 3))
    est = mellon.FunctionEstimator(cov_func=cov, n_landmarks=m, sigma=0.1, predictor_with_uncertainty=True)
    est.fit(x, y)
    return float(est.predict.uncertainty(x[:10]).mean())
attempt("FunctionEstimator uncertainty", fe_unc)

def tsde():
    t = np.repeat(np.arange(3.0), n // 3)
    est = mellon.TimeSensitiveDensityEstimator(cov_func=Cauchy(ls, active_dims=slice(0, d)) * mellon.Matern52(1.0, active_dims=d),
                                               n_landmarks=m)
    return float(est.fit_predict(x, t).mean())
attempt("TimeSensitiveDensityEstimator user x builtin", tsde)

def grad():
    est = mellon.DensityEstimator(cov_func=cov, n_landmarks=m)
    est.fit(x)
    return est.predict.gradient(x[:5]).shape
attempt("Predictor.gradient (expected NotImplementedError)", grad)

def js():
    est = mellon.DensityEstimator(cov_func=cov, n_landmarks=m)
    est.fit(x)
    return len(est.predict.to_json())
attempt("Predictor.to_json", js)
