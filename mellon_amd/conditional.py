"""Predictor builders (mellon/conditional.py): the weight solves run on the device.

  FullConditional               weights = L^-T L^-1 (y - mu),  L = chol(K(x,x) + s I)     :183-264
  LandmarksConditional          `_sparse_solve` with A = Lp^-1 K(xu, x)                    :455-547, :57-66
  LandmarksConditionalCholesky  weights = Lp^-T z                                          :750-818
Only scalar sigma is supported (per-feature / per-observation sigma: S8f).
"""
import logging

import numpy as np

from . import _lib
from .base_predictor import ExpPredictor, Predictor, PredictorTime
from .decomposition import FactorL, FactorLp
from .util import DEFAULT_JITTER, ensure_2d

DEFAULT_SIGMA = 0
logger = logging.getLogger("mellon")


def _scalar_sigma(sigma):
    if sigma is None:
        return None
    s = np.asarray(sigma, dtype=np.float64)
    if s.ndim != 0:
        raise NotImplementedError("Only scalar sigma is supported by the accelerated predictors "
                                  "(per-feature / per-observation sigma: SURVEY.md S8f).")
    return float(s)


def _reject_extras(with_uncertainty, obs_variance):
    pass


def _hc3(residual, h):
    """Corrected squared residuals r^2 / (1 - h)^2 (conditional.py:330-333,607-610)."""
    if residual.ndim > h.ndim:
        h = h[..., None]
    return residual ** 2 / (1 - h) ** 2


def _full_leverage(x, cov_func, sigma, jitter):
    """Training leverage of the full GP, h = 1 - sigma^2 diag((K + sigma^2 I + jitter I)^-1)
    (conditional.py:373-403).  With s = sigma^2 + jitter and L L^T = K + s I,
    k_i^T (K + s I)^-1 k_i = k_ii - s + s^2 A_ii, so A_ii follows from the device's predictive-variance
    kernel: c_i = k_ii - |L^-1 k_i|^2 = s - s^2 A_ii."""
    ctx = _lib.default_context()
    s2 = float(sigma) ** 2
    s = s2 + jitter
    desc = cov_func.lower(x.shape[1])
    fit = ctx.fit_prepare(desc, x, None, s)
    c = ctx.predict_covariance(desc, x, x, fit.Lp(), diag=True)
    return 1.0 - s2 * (s - c) / (s * s), fit


def _landmarks_leverage(Xnew, xu, cov_func, sigma, jitter, L=None):
    """diag(B M^-1 B^T), B = cov(Xnew, xu), M = sigma^2 K_uu + B^T B + jitter I (conditional.py:660-685);
    K_uu = L L^T when the predictor carries L, cov(xu, xu) otherwise -- as in the reference."""
    ctx = _lib.default_context()
    desc = cov_func.lower(xu.shape[1])
    S = ctx.kernel_gram(desc, Xnew, xu)
    K_uu = (np.asarray(L) @ np.asarray(L).T) if L is not None else ctx.kernel_matrix(desc, xu, xu)
    M = float(sigma) ** 2 * K_uu + S
    Lm = ctx.chol_lower(M, add_diag=jitter, jitter=jitter)
    c = ctx.predict_covariance(desc, Xnew, xu, Lm, diag=True)      # k_ii - |Lm^-1 k_u(x_i)|^2
    return cov_func.diag(Xnew) - c


def _parameter_std(sigma, m):
    """Standard deviations of the m parameters as a vector (conditional.py:856-862: `diagonal(sigma)`
    for a vector, `eye(m) * sigma` for a scalar)."""
    s = np.asarray(0.0 if sigma is None else sigma, dtype=np.float64)
    if s.ndim == 0:
        return np.full(m, float(s))
    if s.shape != (m,):
        raise ValueError(f"pre_transformation_std has shape {s.shape}, expected {(m,)}")
    return s


def _attach_uncertainty(pred, Lf, std):
    """with_uncertainty state: L (the factor) and W = L^-T diag(std)
    (conditional.py:330-362 with y_cov_factor = L diag(std), and :853-867)."""
    Lh = np.ascontiguousarray(np.asarray(Lf, dtype=np.float64))
    pred.L = Lh
    pred.W = _lib.default_context().trsm_lower(Lh, np.diag(std), trans=True)
    pred._state_variables |= {"L", "W"}


class _FullConditional:
    _center_name = "x"

    def _leverage(self, Xnew, sigma):      # the training leverage, whatever Xnew (conditional.py:373-403)
        return _full_leverage(self.x, self.cov_func, _scalar_sigma(sigma), self.jitter)[0]

    def __init__(self, x, y, mu, cov_func, L=None, sigma=DEFAULT_SIGMA, jitter=DEFAULT_JITTER, y_cov_factor=None,
                 y_is_mean=False, with_uncertainty=False, obs_variance=False, parameter_std=None):
        _reject_extras(with_uncertainty, obs_variance)
        x = np.ascontiguousarray(ensure_2d(x), dtype=np.float64)
        ctx = _lib.default_context()
        if isinstance(L, (FactorLp, FactorL)) and L.fit.m == x.shape[0] and L.fit.handle is not None \
                and getattr(L.fit, "_has_lp", True):
            fit = L.fit
        elif L is not None:
            Lh = np.asarray(L, dtype=np.float64)
            fit = _lib.Fit.from_L(ctx, Lh, Lp=Lh)
        else:
            logger.info("Recomputing covariance decomposition for predictive function.")
            if y_is_mean:
                diag = jitter                                            # _get_L(x, cov, jitter)
            else:
                s = _scalar_sigma(sigma)
                if s is None and y_cov_factor is None:
                    raise ValueError("No input uncertainty specified. Make sure to set `sigma` or "
                                     "`pre_transformation_std` to quantify uncertainty of the prediction.")
                if y_cov_factor is not None:
                    raise NotImplementedError("y_cov_factor is outside the accelerated path.")
                diag = max(s * s, jitter)                                # add_variance, util.py:296-331
            fit = ctx.fit_prepare(cov_func.lower(x.shape[1]), x, None, diag)
        weights = fit.weights_full(np.asarray(y, dtype=np.float64), mu)  # conditional.py:263-264
        Predictor.__init__(self, cov_func, x, weights, mu, n_obs=x.shape[0], jitter=jitter, sigma=sigma)
        if obs_variance:      # conditional.py:305-362 (scalar sigma): smoothed HC3 observation variance
            s_ = _scalar_sigma(sigma)
            if s_ is None:
                raise ValueError("obs_variance needs the noise level `sigma`.")
            h, fit_v = _full_leverage(x, cov_func, s_, jitter)
            self._corrected_r2 = _hc3(np.asarray(y, dtype=np.float64) - self._mean(x), h)
            self.variance_mu = 0.0
            self.variance_weights = fit_v.weights_full(self._corrected_r2, self.variance_mu)
            self._state_variables |= {"variance_weights", "variance_mu"}
        if with_uncertainty and parameter_std is not None:
            # y_cov_factor = L diag(std) (inference.compute_parameter_cov_factor, inference.py:357-372)
            _attach_uncertainty(self, fit.Lp(), _parameter_std(parameter_std, x.shape[0]))
        elif with_uncertainty:
            # noisy observations (conditional.py:285-304): L = chol(K + sigma^2 I) and
            # W = L^-T L^-1 y_cov_factor with y_cov_factor = sigma I
            s_ = _scalar_sigma(sigma)
            if s_ is None:
                raise ValueError("No input uncertainty specified. Make sure to set `sigma` or "
                                 "`pre_transformation_std` to quantify uncertainty of the prediction.")
            Lh = fit.Lp()
            ctx_ = _lib.default_context()
            self.L = Lh
            self.W = ctx_.trsm_lower(Lh, ctx_.trsm_lower(Lh, np.eye(x.shape[0]) * s_), trans=True)
            self._state_variables |= {"L", "W"}


class _LandmarksConditional:
    _center_name = "landmarks"

    def _leverage(self, Xnew, sigma):
        return _landmarks_leverage(Xnew, self.landmarks, self.cov_func, _scalar_sigma(sigma), self.jitter,
                                   L=getattr(self, "L", None))

    def __init__(self, x, xu, y, mu, cov_func, L=None, Lp=None, sigma=DEFAULT_SIGMA, jitter=DEFAULT_JITTER,
                 y_cov_factor=None, y_is_mean=False, with_uncertainty=False, obs_variance=False):
        _reject_extras(with_uncertainty, obs_variance)
        if with_uncertainty and y_is_mean:
            raise NotImplementedError("with_uncertainty for a landmark conditional on a mean (needs y_cov_factor, "
                                      "conditional.py:579-587) is outside the accelerated path.")
        # y_is_mean (conditional.py:536-537) feeds _sparse_solve with (r, A) unscaled, which is what
        # _process_sigma produces for sigma = 1: L_B L_B^T = A A^T + I, c = L_B^-1 A r.
        s = 1.0 if y_is_mean else _scalar_sigma(sigma)
        if s is None or not s > 0:
            raise ValueError("sigma must be a positive scalar for the landmark conditional "
                             "(the reference divides by sigma^2, conditional.py:157-159).")
        xh = x if isinstance(x, _lib.DeviceArray) else np.ascontiguousarray(ensure_2d(x), dtype=np.float64)
        xu = np.ascontiguousarray(ensure_2d(xu), dtype=np.float64)
        out = _lib.default_context().sparse_solve(cov_func.lower(xu.shape[1]), xh, xu,
                                                  np.asarray(y, dtype=np.float64), mu, s, jitter,
                                                  return_factors=bool(with_uncertainty))
        weights = out[0] if with_uncertainty else out
        Predictor.__init__(self, cov_func, xu, weights, mu, n_obs=xh.shape[0], jitter=jitter, sigma=sigma)
        if obs_variance:          # conditional.py:589-645 (scalar sigma); the leverage here uses K_uu = Lp Lp^T
            ctx = _lib.default_context()
            desc = cov_func.lower(xu.shape[1])
            Lp_h = out[1] if with_uncertainty else ctx.chol_lower(ctx.kernel_matrix(desc, xu, xu), add_diag=jitter,
                                                                  jitter=jitter)
            xfull = xh.to_host() if isinstance(xh, _lib.DeviceArray) else xh
            h = _landmarks_leverage(xfull, xu, cov_func, s, jitter, L=Lp_h)
            self._corrected_r2 = _hc3(np.asarray(y, dtype=np.float64) - self._mean(xfull), h)
            self.variance_mu = 0.0
            self.variance_weights = ctx.sparse_solve(desc, xh, xu, self._corrected_r2, self.variance_mu, s, jitter)
            self._state_variables |= {"variance_weights", "variance_mu"}
        if with_uncertainty:      # conditional.py:571-577: L = Lp, Cs = Lp L_B
            self.L, self.Cs = out[1], out[2]
            self._state_variables |= {"L", "Cs"}


class _LandmarksConditionalCholesky:
    _center_name = "landmarks"

    def __init__(self, xu, pre_transformation, mu, cov_func, n_obs, L=None, sigma=DEFAULT_SIGMA,
                 jitter=DEFAULT_JITTER, y_is_mean=False, with_uncertainty=False, obs_variance=False,
                 obs_x=None, obs_y=None):
        _reject_extras(with_uncertainty, obs_variance)
        xu = np.ascontiguousarray(ensure_2d(xu), dtype=np.float64)
        z = np.asarray(pre_transformation, dtype=np.float64)
        if isinstance(L, (FactorLp, FactorL)) and L.fit.handle is not None:
            weights = L.fit.weights_cholesky(z)                          # conditional.py:818
        else:
            ctx = _lib.default_context()
            if L is None:
                logger.info("Recomputing covariance decomposition for predictive function.")
                Lh = ctx.chol_lower(cov_func(xu, xu), add_diag=jitter, jitter=jitter)
            else:
                Lh = np.asarray(L, dtype=np.float64)
            weights = ctx.trsm_lower(Lh, z, trans=True)
        Predictor.__init__(self, cov_func, xu, weights, mu, n_obs=n_obs, jitter=jitter, sigma=sigma)
        if with_uncertainty:
            Lf = L.fit.Lp() if isinstance(L, (FactorLp, FactorL)) else Lh
            _attach_uncertainty(self, Lf, _parameter_std(sigma, xu.shape[0]))


class FullConditional(_FullConditional, Predictor):
    pass


class ExpFullConditional(_FullConditional, ExpPredictor):
    pass


class FullConditionalTime(_FullConditional, PredictorTime):
    pass


class LandmarksConditional(_LandmarksConditional, Predictor):
    pass


class ExpLandmarksConditional(_LandmarksConditional, ExpPredictor):
    pass


class LandmarksConditionalTime(_LandmarksConditional, PredictorTime):
    pass


class LandmarksConditionalCholesky(_LandmarksConditionalCholesky, Predictor):
    pass


class ExpLandmarksConditionalCholesky(_LandmarksConditionalCholesky, ExpPredictor):
    pass


class LandmarksConditionalCholeskyTime(_LandmarksConditionalCholesky, PredictorTime):
    pass
