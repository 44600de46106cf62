"""Predictor API (mellon/base_predictor.py): `mean` / `__call__` on the device, plus the
reference's JSON wire format for the predictor state (base_predictor.py:541-734) so that
predictors fitted here load in upstream Mellon and vice versa.  Covariance / gradient /
uncertainty methods are outside the accelerated path and raise."""
import bz2
import gzip
import json
import logging
import sys
from datetime import datetime
from importlib import import_module
from math import log

import numpy as np

from . import _lib
from .base_cov import Covariance
from .util import deserialize, ensure_2d, make_multi_time_argument, make_serializable
from .validation import validate_array, validate_bool, validate_time_x

logger = logging.getLogger("mellon")
_REF_MODULE = "mellon.conditional"


class Predictor:
    """Mean function of a conditioned GP: mean(x) = mu + K(x, centers) @ weights."""

    n_obs = None
    d = None
    d_method = None
    _center_name = "landmarks"

    def __init__(self, cov_func, centers, weights, mu, n_obs=None, jitter=None, sigma=None):
        self.cov_func = cov_func
        setattr(self, self._center_name, np.ascontiguousarray(ensure_2d(centers), dtype=np.float64))
        self.weights = np.ascontiguousarray(weights, dtype=np.float64)
        self.mu = float(mu)
        self.jitter = jitter
        self.sigma = sigma
        self.per_feature_sigma = False
        self.n_input_features = getattr(self, self._center_name).shape[1]
        self.n_obs = n_obs
        self._state_variables = {self._center_name, "weights", "mu", "jitter", "sigma", "per_feature_sigma"}

    @property
    def centers(self):
        return getattr(self, self._center_name)

    def __repr__(self):
        return (f'A predictor of class "{self.__class__.__name__}" with covariance function '
                f'"{self.cov_func}" and data:\n' +
                "\n".join(f"    {k}: {np.shape(getattr(self, k))}" for k in sorted(self._state_variables)))

    # -- evaluation (conditional.py:366-373,651-658,899-906 through base_predictor.py:180-257) ------
    def _mean(self, Xnew, out=None):
        ctx = _lib.default_context()
        return ctx.predict_mean(self.cov_func.lower(self.n_input_features), Xnew, self.centers,
                                self.weights, self.mu, out=out)

    def mean(self, x, normalize=False, out=None):
        """reference base_predictor.py:180-257.  `out` (beyond the reference's signature): a _lib.DeviceArray that
        receives the predictions in HBM -- with HBM-resident queries a batched predict then moves nothing over PCIe."""
        x = validate_array(x, "x")
        if not isinstance(x, _lib.DeviceArray):
            x = np.ascontiguousarray(ensure_2d(x), dtype=np.float64)
        normalize = validate_bool(normalize, "normalize")
        if x.shape[1] != self.n_input_features:
            raise ValueError(
                f"The predictor was trained on data with {self.n_input_features} features. "
                f"However, the provided input data has {x.shape[1]} features. "
                "Please ensure that the input data has the same number of features as the training data.")
        if normalize:
            if self.n_obs is None or self.n_obs == 0:
                raise ValueError("Cannot normalize without n_obs. Please set self.n_obs to the number "
                                 "of samples/cells trained on to enable normalization.")
            if out is not None:
                raise ValueError("normalize=True is not combined with a device-resident output")
            return self._mean(x) - log(self.n_obs)
        return self._mean(x, out=out)

    __call__ = mean

    def gradient(self, x, jit=True):
        """Gradient of the predicted mean at each row of x, shape == x.shape (base_predictor.py:490-505).
        The reference differentiates `_mean` with jax.jacrev; here the analytic kernel gradient is
        contracted with the weights on the device.  `jit` is accepted and ignored."""
        x = self._check_features(x)
        ctx, desc = _lib.default_context(), self.cov_func.lower(self.n_input_features)
        if self.weights.ndim == 2:      # p outputs: (n, p, d), the layout derivatives.gradient gives (derivatives.py:76-80)
            return np.stack([ctx.predict_gradient(desc, x, self.centers, np.ascontiguousarray(self.weights[:, c]))
                             for c in range(self.weights.shape[1])], axis=1)
        return ctx.predict_gradient(desc, x, self.centers, self.weights)

    # -- predictive uncertainty (base_predictor.py:330-428; conditional.py _covariance / _mean_covariance) --
    def _check_features(self, x):
        x = validate_array(x, "x")
        x = np.ascontiguousarray(ensure_2d(x), dtype=np.float64)
        if x.shape[1] != self.n_input_features:
            raise ValueError(
                f"The predictor was trained on data with {self.n_input_features} features. "
                f"However, the provided input data has {x.shape[1]} features. "
                "Please ensure that the input data has the same number of features as the training data.")
        return x

    def _has_per_feature_sigma(self):
        """Whether the predictor was fitted with a sigma per output (base_predictor.py:360-362)."""
        return bool(getattr(self, "per_feature_sigma", False))

    def covariance(self, x, diag=True, noise_free=False):
        """k(x,x) - A A^T with A = cov(x, centers) L^-T (variances when diag=True).  A predictor fitted with a
        per-feature sigma only has the noise-free covariance and asks for `noise_free=True`
        (base_predictor.py:364-420)."""
        if self._has_per_feature_sigma() and not noise_free:
            raise ValueError(
                "This predictor was fitted with per-feature sigma, so the "
                "covariance is noise-free (sigma=0) and does not include "
                "observation noise. Pass noise_free=True to acknowledge this "
                "and obtain the noise-free covariance, then account for "
                "observation noise separately (e.g., via obs_variance).")
        x = self._check_features(x)
        if not hasattr(self, "L") or self.L is None:
            raise ValueError("The predictor was computed without covariance. "
                             "Recompute setting `with_uncertainty=True.`")
        ctx = _lib.default_context()
        desc = self.cov_func.lower(self.n_input_features)
        cov = ctx.predict_covariance(desc, x, self.centers, np.asarray(self.L), diag=diag)
        Cs = getattr(self, "Cs", None)
        if Cs is not None:
            # noisy landmark conditional (conditional.py:707-716): + |Cs^-1 K_ux|^2, i.e. k - (k - |Cs^-1 K|^2)
            kss = self.cov_func.diag(x) if diag else self.cov_func(x, x)
            cov = cov + (kss - ctx.predict_covariance(desc, x, self.centers, np.asarray(Cs), diag=diag))
        return cov

    def mean_covariance(self, x, diag=True):
        """(K W)(K W)^T: uncertainty of the mean inherited from the parameter uncertainty W."""
        x = self._check_features(x)
        if not hasattr(self, "W") or self.W is None:
            raise ValueError(
                "The predictor was computed without uncertainty, e.g., using ADVI. "
                "Recompute setting `with_uncertainty=True.` and define `pre_transformation_std`"
                ", e.g., by using `optimizer='advi'`.")
        return _lib.default_context().predict_mean_covariance(self.cov_func.lower(self.n_input_features), x,
                                                              self.centers, np.asarray(self.W), diag=diag)

    def uncertainty(self, x, diag=True):
        """covariance + mean_covariance (base_predictor.py:390-428)."""
        return self.covariance(x, diag=diag) + self.mean_covariance(x, diag=diag)

    def hessian(self, x, jit=True):
        """Hessian of the predicted mean at each row of x, shape x.shape + (d,) (base_predictor.py:507-521).  The
        reference applies jacfwd(jacrev(.)); here the closed-form second derivatives of the kernels are contracted
        with the weights on the device.  `jit` is accepted and ignored."""
        x = self._check_features(x)
        ctx, desc = _lib.default_context(), self.cov_func.lower(self.n_input_features)
        if self.weights.ndim == 2:      # p outputs: (n, p, d, d) (derivatives.py:114-117)
            return np.stack([ctx.predict_hessian(desc, x, self.centers, np.ascontiguousarray(self.weights[:, c]))
                             for c in range(self.weights.shape[1])], axis=1)
        return ctx.predict_hessian(desc, x, self.centers, self.weights)

    def hessian_log_determinant(self, x, jit=True):
        """(signs, log |det|) of the Hessian at each row of x (base_predictor.py:523-539)."""
        sign, logdet = np.linalg.slogdet(self.hessian(x))
        return sign, logdet

    # -- leverage / observation variance (base_predictor.py:263-355) -------------------------------------------
    def leverage(self, x):
        """Hat-matrix diagonal h_i used by the HC3 correction r^2 / (1 - h)^2."""
        x = self._check_features(x)
        if not hasattr(self, "_leverage"):
            raise NotImplementedError(f"{self.__class__.__name__} has no leverage.")
        if self.sigma is None:
            raise ValueError("leverage needs the noise level `sigma` of the fit.")
        return self._leverage(x, self.sigma)

    def loo_residuals_squared(self, x, y):
        """Squared leave-one-out residuals through the leverage shortcut (base_predictor.py:290-324)."""
        x = self._check_features(x)
        y = np.asarray(validate_array(y, "y"), dtype=np.float64)
        residual = y - self._mean(x)
        h = self.leverage(x)
        if residual.ndim > h.ndim:
            h = h[..., None]
        return residual ** 2 / (1 - h) ** 2

    def obs_variance(self, x):
        """Smoothed observation variance: a second GP on the HC3-corrected squared residuals."""
        x = self._check_features(x)
        if getattr(self, "variance_weights", None) is None:
            raise ValueError("The predictor was computed without obs_variance. "
                             "Recompute setting `obs_variance=True`.")
        ctx = _lib.default_context()
        return ctx.predict_mean(self.cov_func.lower(self.n_input_features), x, self.centers,
                                np.asarray(self.variance_weights), float(self.variance_mu))

    # -- serialization --------------------------------------------------------------------------------
    def _data_dict(self):
        return {k: getattr(self, k) for k in self._state_variables}

    def __getstate__(self):
        from . import __version__
        data = self._data_dict()
        data.update({"n_input_features": self.n_input_features, "n_obs": self.n_obs, "d": self.d,
                     "d_method": self.d_method, "_state_variables": self._state_variables})
        return {
            "data": {k: make_serializable(v) for k, v in data.items()},
            "cov_func": self.cov_func.__getstate__(),
            "metadata": {"classname": self.__class__.__name__, "module_name": _REF_MODULE,
                         "module_version": __version__, "serialization_date": datetime.now().isoformat(),
                         "python_version": sys.version},
        }

    def __setstate__(self, state):
        for name, value in state["data"].items():
            setattr(self, name, deserialize(value))
        # files written by mellon 1.3.1 carry neither `_state_variables` nor `n_obs` (tests/test_density_estimator.py:
        # 139-151): every other entry of "data" is then a state variable, and n_obs stays None (normalize=True raises)
        meta = {"n_input_features", "n_obs", "d", "d_method", "_state_variables"}
        self._state_variables = set(getattr(self, "_state_variables", None) or (set(state["data"]) - meta))
        self.cov_func = Covariance.from_dict(state["cov_func"])
        for k in (self._center_name, "weights", "L", "W"):
            if getattr(self, k, None) is not None:
                setattr(self, k, np.ascontiguousarray(getattr(self, k), dtype=np.float64))

    def copy(self):
        new = self.__class__.__new__(self.__class__)
        new.__setstate__(self.__getstate__())
        return new

    def to_dict(self):
        return self.__getstate__()

    def to_json(self, filename=None, compress=None):
        json_str = json.dumps(self.to_dict())
        if filename is None:
            return json_str
        if compress == "gzip":
            filename = filename if str(filename).endswith(".gz") else str(filename) + ".gz"
            with gzip.open(filename, "wt") as f:
                f.write(json_str)
        elif compress == "bz2":
            filename = filename if str(filename).endswith(".bz2") else str(filename) + ".bz2"
            with bz2.open(filename, "wt") as f:
                f.write(json_str)
        elif compress is None:
            with open(filename, "w") as f:
                f.write(json_str)
        else:
            raise ValueError(f'Unknown compression format {compress}.\nAvailabe formats are "gzip", "bz2" and None.')
        logger.info(f"Written predictor to {filename}.")

    @classmethod
    def from_dict(cls, data_dict):
        clsname = data_dict["metadata"]["classname"]
        from . import conditional
        Sub = getattr(conditional, clsname, None)
        if Sub is None:
            Sub = getattr(import_module(data_dict["metadata"]["module_name"]), clsname)
        inst = Sub.__new__(Sub)
        inst.__setstate__(data_dict)
        return inst

    @classmethod
    def from_json_str(cls, json_str):
        return cls.from_dict(json.loads(json_str))

    @classmethod
    def from_json(cls, filepath, compress=None):
        filename = str(filepath)
        if compress == "gzip" or filename.endswith(".gz"):
            opener = gzip.open
        elif compress == "bz2" or filename.endswith(".bz2"):
            opener = bz2.open
        else:
            opener = open
        with opener(filepath, "rt") as f:
            return cls.from_json_str(f.read())


class ExpPredictor(Predictor):
    """exp of the mean (reference base_predictor.py ExpPredictor)."""

    def mean(self, x, logscale=False):
        """exp(mean), or the mean itself with `logscale=True` (base_predictor.py:748-787)."""
        logscale = validate_bool(logscale, "logscale")
        log_value = Predictor.mean(self, x)
        return log_value if logscale else np.exp(log_value)

    __call__ = mean

    def gradient(self, x, jit=True):
        return np.exp(Predictor.mean(self, x))[:, None] * Predictor.gradient(self, x)

    def hessian(self, x, jit=True):
        """Hessian of exp(f): e^f (H_f + grad f grad f^T)."""
        g = Predictor.gradient(self, x)
        return np.exp(Predictor.mean(self, x))[:, None, None] * (Predictor.hessian(self, x) + g[:, :, None] * g[:, None, :])


class PredictorTime(Predictor):
    """Predictor whose last input column is time (reference base_predictor.py:872-948)."""

    @make_multi_time_argument
    def mean(self, Xnew, time=None, normalize=False):
        Xnew = validate_array(Xnew, "Xnew")
        if isinstance(Xnew, _lib.DeviceArray) and time is None:
            return Predictor.mean(self, Xnew, normalize=normalize)      # HBM-resident queries [state | time]: used in place
        Xnew = np.ascontiguousarray(ensure_2d(Xnew), dtype=np.float64)
        x = validate_time_x(Xnew, time, n_features=self.n_input_features, cast_scalar=True)
        return Predictor.mean(self, np.ascontiguousarray(x), normalize=normalize)

    __call__ = mean

    @make_multi_time_argument
    def gradient(self, x, time=None, jit=True):
        """Gradient with respect to the state columns at the given time(s) (base_predictor.py:1094-1124)."""
        return Predictor.gradient(self, self._with_time(x, time))[:, :-1]

    @make_multi_time_argument
    def hessian(self, x, time=None, jit=True):
        """Hessian with respect to the state columns at the given time(s) (base_predictor.py:1127-1159)."""
        return Predictor.hessian(self, self._with_time(x, time))[:, :-1, :-1]

    @make_multi_time_argument
    def hessian_log_determinant(self, x, time=None, jit=True):
        """base_predictor.py:1162-1194."""
        sign, logdet = np.linalg.slogdet(Predictor.hessian(self, self._with_time(x, time))[:, :-1, :-1])
        return sign, logdet

    @make_multi_time_argument
    def time_derivative(self, x, time=None, jit=True):
        """Derivative with respect to time (base_predictor.py:1052-1091)."""
        return Predictor.gradient(self, self._with_time(x, time))[:, -1]

    def _with_time(self, Xnew, time):
        Xnew = np.ascontiguousarray(ensure_2d(validate_array(Xnew, "Xnew")), dtype=np.float64)
        return np.ascontiguousarray(validate_time_x(Xnew, time, n_features=self.n_input_features, cast_scalar=True))

    @make_multi_time_argument
    def covariance(self, Xnew, time=None, diag=True, noise_free=False):
        return Predictor.covariance(self, self._with_time(Xnew, time), diag=diag, noise_free=noise_free)

    @make_multi_time_argument
    def mean_covariance(self, Xnew, time=None, diag=True):
        return Predictor.mean_covariance(self, self._with_time(Xnew, time), diag=diag)

    @make_multi_time_argument
    def uncertainty(self, Xnew, time=None, diag=True):
        x = self._with_time(Xnew, time)
        return Predictor.covariance(self, x, diag=diag) + Predictor.mean_covariance(self, x, diag=diag)
