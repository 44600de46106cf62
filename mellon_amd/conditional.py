"""Predictor builders (mellon/conditional.py): the weight solves run on the device.

  FullConditional               weights = L^-T L^-1 (y - mu),  L = chol(K(x,x) + s I)     :183-264
  LandmarksConditional          `_sparse_solve` with A = Lp^-1 K(xu, x)                    :455-547, :57-66
  LandmarksConditionalCholesky  weights = Lp^-T z                                          :750-818
Noise models (conditional.py:13-43,100-159): one scalar sigma; one sigma per output column ("per-gene",
shape (p,) or (1, p)); one per cell (shape (n,), element-wise std) and one per cell and output ((n, p)).
A full covariance matrix as sigma and `y_cov_factor` stay outside the accelerated path.
"""
import logging

import numpy as np

from . import _lib
from .base_predictor import ExpPredictor, Predictor, PredictorTime
from .decomposition import FactorL, FactorLp
from .util import DEFAULT_JITTER, ensure_2d

DEFAULT_SIGMA = 0
SPECTRAL_MIN_LEVELS = 8      # distinct noise levels above which the landmark leverage goes spectral
logger = logging.getLogger("mellon")


def _scalar_sigma(sigma):
    if sigma is None:
        return None
    s = np.asarray(sigma, dtype=np.float64)
    if s.ndim != 0:
        raise NotImplementedError("This code path takes one scalar sigma.")
    return float(s)


def _is_per_feature_sigma(sigma, y):
    """One sigma per output column of a 2-D y -- shapes (p,), (1, p), (n, p) (conditional.py:13-36)."""
    if sigma is None or np.ndim(sigma) == 0:
        return False
    sigma, y = np.asarray(sigma), np.asarray(y)
    if sigma.ndim == 2 and sigma.shape[0] == 1 and y.ndim == 2 and sigma.shape[1] == y.shape[1]:
        return True
    if sigma.ndim == 2 and y.ndim == 2 and sigma.shape == y.shape:
        return True
    if sigma.ndim == 1 and y.ndim == 2 and sigma.shape[0] == y.shape[1]:
        if sigma.shape[0] == y.shape[0]:
            logger.warning(f"sigma length {sigma.shape[0]} matches both n_obs and n_features. "
                           "Interpreting as per-feature. Pass sigma with shape (n, 1) for per-observation.")
        return True
    return False


def _normalize_per_feature_sigma(sigma):
    """(1, p) -> (p,) (conditional.py:39-43)."""
    sigma = np.asarray(sigma, dtype=np.float64)
    return sigma[0] if sigma.ndim == 2 and sigma.shape[0] == 1 else sigma


def _per_output_levels(sigma):
    """The distinct noise levels of a per-output sigma and, for each, the output columns that carry it."""
    sig = _normalize_per_feature_sigma(sigma)
    if sig.ndim != 1:
        raise NotImplementedError("leverage / obs_variance take one sigma per output column; the reference's "
                                  "(n, p) sigma only defines the weights (conditional.py:313-323 needs a scalar "
                                  "per column).")
    levels, inverse = np.unique(sig, return_inverse=True)
    return sig, [(float(v), np.flatnonzero(inverse == k)) for k, v in enumerate(levels)]


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


def _full_leverage_any(x, cov_func, sigma, jitter):
    """Scalar sigma -> (n,); per-output sigma -> (n, p), one column per output (conditional.py:385-403)."""
    if np.ndim(sigma) == 0:
        return _full_leverage(x, cov_func, float(sigma), jitter)[0]
    sig, levels = _per_output_levels(sigma)
    if len(levels) > SPECTRAL_MIN_LEVELS:
        ctx = _lib.default_context()
        return ctx.full_conditional_noise(cov_func.lower(x.shape[1]), x, np.zeros((x.shape[0], sig.shape[0])), 0.0,
                                          sig, jitter, leverage=True)[1]
    h = np.empty((x.shape[0], sig.shape[0]))
    for value, cols in levels:
        h[:, cols] = _full_leverage(x, cov_func, value, jitter)[0][:, None]
    return h


def _landmarks_leverage_any(Xnew, xu, cov_func, sigma, jitter, L=None):
    """Scalar sigma -> (n,); per-output sigma -> (n, p) (conditional.py:660-685).  B^T B and K_uu are formed
    once; each distinct noise level costs one m x m Cholesky and one pass over the cells."""
    if np.ndim(sigma) == 0:
        return _landmarks_leverage(Xnew, xu, cov_func, float(sigma), jitter, L=L)
    sig, levels = _per_output_levels(sigma)
    ctx = _lib.default_context()
    desc = cov_func.lower(xu.shape[1])
    if L is not None and len(levels) > SPECTRAL_MIN_LEVELS:
        # many levels and K_uu = L L^T with a known factor: one eigendecomposition for all of them
        per_level = ctx.landmark_leverage(desc, Xnew, xu, np.asarray(L), [v for v, _ in levels], jitter)
        h = np.empty((Xnew.shape[0], sig.shape[0]))
        for k, (_, cols) in enumerate(levels):
            h[:, cols] = per_level[:, k:k + 1]
        return h
    S = ctx.kernel_gram(desc, Xnew, xu)
    K_uu = (np.asarray(L) @ np.asarray(L).T) if L is not None else ctx.kernel_matrix(desc, xu, xu)
    kdiag = cov_func.diag(Xnew)
    h = np.empty((Xnew.shape[0], sig.shape[0]))
    for value, cols in levels:
        Lm = ctx.chol_lower(value ** 2 * K_uu + S, add_diag=jitter, jitter=jitter)
        h[:, cols] = (kdiag - ctx.predict_covariance(desc, Xnew, xu, Lm, diag=True))[:, None]
    return h


def _sparse_solve_per_output(ctx, desc, x, xu, y, mu, sigma, jitter):
    """`_sparse_solve` for every output column with its own sigma (conditional.py:526-545).  Inside
    mln_sparse_solve_noise adjacent columns with one level share an L_B; beyond 32 runs one eigendecomposition of
    A A^T serves every level."""
    sig = _normalize_per_feature_sigma(sigma)
    y = np.asarray(y, dtype=np.float64)
    if sig.ndim == 2:       # (n, p): an element-wise noise vector per output, one pass per output
        return np.stack([ctx.sparse_solve_noise(desc, x, xu, np.ascontiguousarray(y[:, g]), mu,
                                                np.ascontiguousarray(sig[:, g]), ctx.SIGMA_PER_CELL, jitter)
                         for g in range(y.shape[1])], axis=1)
    return ctx.sparse_solve_noise(desc, x, xu, y, mu, sig, ctx.SIGMA_PER_OUTPUT, jitter)


def _sigma_to_y_cov_factor(sigma, y_cov_factor, n):
    """conditional.py:101-135: the left factor of the observation noise from `sigma`, or the caller's own `y_cov_factor`
    (one or the other); pinned by the reference's tests/test_sigma_to_y_cov_factor.py:6-43."""
    if sigma is None and y_cov_factor is None:
        raise ValueError("No input uncertainty specified. Make sure to set `sigma` or `pre_transformation_std`, "
                         'e.g., by using `optimizer="advi", to quantify uncertainty of the prediction.')
    if y_cov_factor is not None and sigma is not None and np.any(np.asarray(sigma) > 0):
        raise ValueError("One can specify either `sigma` or `y_cov_factor` to describe input noise, but not both.")
    if y_cov_factor is not None:
        return np.asarray(y_cov_factor, dtype=np.float64)
    sig = np.asarray(sigma, dtype=np.float64)
    if sig.ndim == 0:
        return np.eye(n) * float(sig)
    if sig.ndim == 1:
        return np.diag(sig)
    out = np.zeros((n,) + sig.shape)              # a leading dimension for the diagonal (conditional.py:122-131)
    out[np.arange(n), np.arange(n), ...] = sig
    return out


def _chol_with_noise_factor(ctx, K, M, jitter):
    """chol(K + M M^T + diag(max(jitter - diag(M M^T), 0)))  (util.add_variance's matrix branch, util.py:326-330, inside
    conditional._get_L, conditional.py:69-81); M M^T accumulates on the matrix cores (mln_gemm), the factorisation is the
    device Cholesky."""
    M = np.ascontiguousarray(M, dtype=np.float64)
    if M.ndim != 2 or M.shape[0] != K.shape[0]:
        raise NotImplementedError("y_cov_factor must be a matrix with one row per training point.")
    Kp = np.array(K, dtype=np.float64)
    ctx.gemm(M, M, tb=True, alpha=1.0, beta=1.0, out=Kp)
    noise_diag = np.einsum("ij,ij->i", M, M)
    Kp[np.diag_indices_from(Kp)] += np.where(noise_diag < jitter, jitter - noise_diag, 0.0)
    return ctx.chol_lower(Kp, jitter=jitter)


def _chol_with_diag(ctx, K, diag_values, jitter):
    M = np.array(K, dtype=np.float64)
    M[np.diag_indices_from(M)] += diag_values
    return ctx.chol_lower(M, jitter=jitter)


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
        return _full_leverage_any(self.x, self.cov_func, sigma, self.jitter)

    def __init__(self, x, y, mu, cov_func, L=None, sigma=DEFAULT_SIGMA, jitter=DEFAULT_JITTER, y_cov_factor=None,
                 y_is_mean=False, with_uncertainty=False, obs_variance=False, parameter_std=None, factor=None):
        x = np.ascontiguousarray(ensure_2d(x), dtype=np.float64)
        ctx = _lib.default_context()
        yh = np.asarray(y, dtype=np.float64)
        n = x.shape[0]
        per_feature = _is_per_feature_sigma(sigma, yh)
        per_cell = (not per_feature and sigma is not None and np.ndim(sigma) == 1 and L is None and not y_is_mean)
        fit = None
        var_state = None
        Lh_ycf = ycf_h = None      # factor and noise factor of the `y_cov_factor=` route
        if per_feature:
            # one solve per output with chol(K + sigma_g^2 I + jitter I) (conditional.py:239-251); outputs that share
            # a noise level share the factorisation, and everything a level needs is done while its factor is resident
            sig = _normalize_per_feature_sigma(sigma)
            desc = cov_func.lower(x.shape[1])
            weights = np.empty((n, yh.shape[1]))
            if sig.ndim == 1 and len(_per_output_levels(sigma)[1]) > SPECTRAL_MIN_LEVELS:
                # many levels: one eigendecomposition of K(x, x) serves them all
                out = ctx.full_conditional_noise(desc, x, yh, mu, sig, jitter, obs_variance=bool(obs_variance))
                if obs_variance:
                    weights, var_state = out[0], (out[2], out[3])
                else:
                    weights = out
            elif sig.ndim == 1:
                if obs_variance:
                    var_state = (np.empty_like(yh), np.empty_like(weights))      # corrected r^2, variance weights
                for value, cols in _per_output_levels(sigma)[1]:
                    if obs_variance:       # conditional.py:305-362, column by column
                        h_v, fit_v = _full_leverage(x, cov_func, value, jitter)
                    else:
                        fit_v = ctx.fit_prepare(desc, x, None, value ** 2 + jitter)
                    y_c = np.ascontiguousarray(yh[:, cols])
                    weights[:, cols] = w_c = fit_v.weights_full(y_c, mu)
                    if obs_variance:
                        cr2 = _hc3(y_c - ctx.predict_mean(desc, x, x, w_c, float(mu)), h_v)
                        var_state[0][:, cols] = cr2
                        var_state[1][:, cols] = fit_v.weights_full(cr2, 0.0)
                    fit_v.close()
            else:
                K = ctx.kernel_matrix(desc, x, x)
                for g in range(yh.shape[1]):
                    Lg = _chol_with_diag(ctx, K, sig[:, g] ** 2 + jitter, jitter)
                    weights[:, g] = ctx.trsm_lower(Lg, ctx.trsm_lower(Lg, yh[:, g] - mu), trans=True)
        elif per_cell:
            # y_cov_factor = diag(sigma): K + diag(max(sigma_i^2, jitter)) (conditional.py:118-119, util.py:326-329)
            sig = np.asarray(sigma, dtype=np.float64)
            if sig.shape != (n,):
                raise ValueError(f"sigma has shape {sig.shape}; expected a scalar, ({n},) or one value per output.")
            K = ctx.kernel_matrix(cov_func.lower(x.shape[1]), x, x)
            Lh = _chol_with_diag(ctx, K, np.maximum(sig ** 2, jitter), jitter)
            weights = ctx.trsm_lower(Lh, ctx.trsm_lower(Lh, yh - mu), trans=True)
        else:
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
                    if y_cov_factor is not None:
                        # the caller's own noise factor: L = chol(K + M M^T [+ diagonal up to the jitter])  (conditional.py:260-262)
                        ycf_h = _sigma_to_y_cov_factor(sigma, y_cov_factor, n)
                        K = ctx.kernel_matrix(cov_func.lower(x.shape[1]), x, x)
                        Lh_ycf = _chol_with_noise_factor(ctx, K, ycf_h, jitter)
                    else:
                        s = _scalar_sigma(sigma)
                        if s is None:
                            raise ValueError("No input uncertainty specified. Make sure to set `sigma` or "
                                             "`pre_transformation_std` to quantify uncertainty of the prediction.")
                        diag = max(s * s, jitter)                            # add_variance, util.py:296-331
                if Lh_ycf is None:
                    fit = ctx.fit_prepare(cov_func.lower(x.shape[1]), x, None, diag)
            if Lh_ycf is not None:
                weights = ctx.trsm_lower(Lh_ycf, ctx.trsm_lower(Lh_ycf, yh - mu), trans=True)
            else:
                weights = fit.weights_full(yh, mu)                           # conditional.py:263-264
        Predictor.__init__(self, cov_func, x, weights, mu, n_obs=x.shape[0], jitter=jitter, sigma=sigma)
        self.per_feature_sigma = bool(per_feature)
        if obs_variance:      # conditional.py:305-362: smoothed HC3 observation variance
            if sigma is None:
                raise ValueError("obs_variance needs the noise level `sigma`.")
            self.variance_mu = 0.0
            if np.ndim(sigma) >= 1:
                if var_state is None:
                    raise NotImplementedError("obs_variance takes a scalar sigma or one sigma per output column.")
                self._corrected_r2, self.variance_weights = var_state
            else:
                h, fit_v = _full_leverage(x, cov_func, float(sigma), jitter)
                self._corrected_r2 = _hc3(yh - self._mean(x), h)
                self.variance_weights = fit_v.weights_full(self._corrected_r2, self.variance_mu)
            self._state_variables |= {"variance_weights", "variance_mu"}
        if with_uncertainty and parameter_std is not None and fit is not None \
                and np.ndim(parameter_std) == 1 and np.shape(parameter_std)[0] != n and factor is not None:
            # low-rank factor (full_nystroem): y_cov_factor = L_lowrank diag(std) is n x rank and
            # W = Lf^-T Lf^-1 y_cov_factor with Lf the recomputed full factor (conditional.py:292-304)
            Lh = fit.Lp()
            ycf = np.asarray(factor, dtype=np.float64) * np.asarray(parameter_std, dtype=np.float64)[None, :]
            self.L = Lh
            self.W = ctx.trsm_lower(Lh, ctx.trsm_lower(Lh, np.ascontiguousarray(ycf)), trans=True)
            self._state_variables |= {"L", "W"}
        elif with_uncertainty and parameter_std is not None:
            # y_cov_factor = L diag(std) (inference.compute_parameter_cov_factor, inference.py:357-372)
            _attach_uncertainty(self, fit.Lp(), _parameter_std(parameter_std, x.shape[0]))
        elif with_uncertainty and per_feature:
            # noise-free covariance, no mean covariance (conditional.py:288-291)
            self.L = ctx.fit_prepare(cov_func.lower(x.shape[1]), x, None, jitter).Lp()
            self._state_variables |= {"L"}
        elif with_uncertainty and Lh_ycf is not None:
            # W = L^-T L^-1 y_cov_factor with the caller's factor (conditional.py:302-306)
            self.L = Lh_ycf
            self.W = ctx.trsm_lower(Lh_ycf, ctx.trsm_lower(Lh_ycf, np.ascontiguousarray(ycf_h)), trans=True)
            self._state_variables |= {"L", "W"}
        elif with_uncertainty:
            # noisy observations (conditional.py:285-304): L = chol(K + sigma^2 I) and
            # W = L^-T L^-1 y_cov_factor with y_cov_factor = sigma I (diag(sigma) for one sigma per cell)
            if sigma is None:
                raise ValueError("No input uncertainty specified. Make sure to set `sigma` or "
                                 "`pre_transformation_std` to quantify uncertainty of the prediction.")
            if not per_cell:
                Lh = fit.Lp()
            self.L = Lh
            ycf = np.diag(np.broadcast_to(np.asarray(sigma, dtype=np.float64), (n,)))
            self.W = ctx.trsm_lower(Lh, ctx.trsm_lower(Lh, ycf), trans=True)
            self._state_variables |= {"L", "W"}


class _LandmarksConditional:
    _center_name = "landmarks"

    def _leverage(self, Xnew, sigma):
        return _landmarks_leverage_any(Xnew, self.landmarks, self.cov_func, sigma, self.jitter,
                                       L=getattr(self, "L", None))

    def __init__(self, x, xu, y, mu, cov_func, L=None, Lp=None, sigma=DEFAULT_SIGMA, jitter=DEFAULT_JITTER,
                 y_cov_factor=None, y_is_mean=False, with_uncertainty=False, obs_variance=False, parameter_std=None):
        if with_uncertainty and y_is_mean and y_cov_factor is None:
            # y_cov_factor = L diag(std): the factor of the covariance of the mean on the training cells
            # (inference.compute_parameter_cov_factor, inference.py:357-372)
            if parameter_std is None or L is None:
                raise ValueError("No input uncertainty specified. Make sure to set `sigma` or "
                                 "`pre_transformation_std` to quantify uncertainty of the prediction.")
            y_cov_factor = np.asarray(L, dtype=np.float64) * np.asarray(parameter_std, dtype=np.float64)[None, :]
        ctx = _lib.default_context()
        xh = x if isinstance(x, _lib.DeviceArray) else np.ascontiguousarray(ensure_2d(x), dtype=np.float64)
        xu = np.ascontiguousarray(ensure_2d(xu), dtype=np.float64)
        y_resident = isinstance(y, _lib.DeviceArray)              # targets already in HBM: scalar-sigma solve only
        yh = y if y_resident else np.asarray(y, dtype=np.float64)
        desc = cov_func.lower(xu.shape[1])
        per_feature = False if y_resident else _is_per_feature_sigma(sigma, yh)
        vector = (not per_feature) and (not y_is_mean) and sigma is not None and np.ndim(sigma) >= 1
        if y_resident and (vector or obs_variance or np.ndim(sigma) > 0):
            raise NotImplementedError("device-resident targets take a scalar sigma and no observation variance")
        Lp_h = Cs_h = None

        def solve(target, mean, want_factors=False):
            """`_sparse_solve` of `target - mean` under this conditional's noise model."""
            if per_feature:
                return _sparse_solve_per_output(ctx, desc, xh, xu, target, mean, sigma, jitter)
            if vector:
                return ctx.sparse_solve_noise(desc, xh, xu, target, mean, sig_cell, ctx.SIGMA_PER_CELL, jitter)
            return ctx.sparse_solve(desc, xh, xu, target, mean, s, jitter, return_factors=want_factors)

        if vector:
            # element-wise standard deviation of the cells (conditional.py:155-159); y must be 1-D there
            sig_cell = np.asarray(sigma, dtype=np.float64)
            if yh.ndim > 1 and sig_cell.shape == yh.shape:
                raise NotImplementedError("FunctionEstimator not implemented for multiple noises.")
            if sig_cell.ndim == 2 and sig_cell.shape == (xh.shape[0], xh.shape[0]):
                raise NotImplementedError("A full covariance matrix as sigma is outside the accelerated path.")
            if sig_cell.shape != yh.shape or yh.ndim != 1:
                raise ValueError("Unsupported sigma configuration.")
            if not np.all(sig_cell > 0):
                raise ValueError("sigma must be positive for the landmark conditional.")
        elif per_feature:
            if not np.all(np.asarray(sigma, dtype=np.float64) > 0):
                raise ValueError("sigma must be positive for the landmark conditional.")
        else:
            # y_is_mean (conditional.py:536-537) feeds _sparse_solve with (r, A) unscaled, which is what
            # _process_sigma produces for sigma = 1: L_B L_B^T = A A^T + I, c = L_B^-1 A r.
            s = 1.0 if y_is_mean else _scalar_sigma(sigma)
            if s is None or not s > 0:
                raise ValueError("sigma must be positive for the landmark conditional "
                                 "(the reference divides by sigma^2, conditional.py:157-159).")
        want_factors = bool(with_uncertainty) and not per_feature and not vector
        out = solve(yh, mu, want_factors)
        if want_factors:
            weights, Lp_h, Cs_h = out
        else:
            weights = out
        Predictor.__init__(self, cov_func, xu, weights, mu, n_obs=xh.shape[0], jitter=jitter, sigma=sigma)
        self.per_feature_sigma = bool(per_feature)
        if (obs_variance or with_uncertainty) and Lp_h is None:
            Lp_h = ctx.chol_lower(ctx.kernel_matrix(desc, xu, xu), add_diag=jitter, jitter=jitter)
        if obs_variance:          # conditional.py:589-645; the leverage here uses K_uu = Lp Lp^T
            if vector or (per_feature and _normalize_per_feature_sigma(sigma).ndim != 1):
                raise NotImplementedError("obs_variance takes a scalar sigma or one sigma per output column.")
            xfull = xh.to_host() if isinstance(xh, _lib.DeviceArray) else xh
            h = _landmarks_leverage_any(xfull, xu, cov_func, 1.0 if y_is_mean else sigma, jitter, L=Lp_h)
            self._corrected_r2 = _hc3(yh - self._mean(xfull), h)
            self.variance_mu = 0.0
            self.variance_weights = solve(self._corrected_r2, self.variance_mu)
            self._state_variables |= {"variance_weights", "variance_mu"}
        if with_uncertainty:      # conditional.py:571-577: L = Lp, Cs = Lp L_B (scalar sigma only)
            self.L = Lp_h
            self._state_variables |= {"L"}
            if Cs_h is not None:
                self.Cs = Cs_h
                self._state_variables |= {"Cs"}
            if y_is_mean:     # conditional.py:579-587: W = Lp^-T L_B^-T L_B^-1 A y_cov_factor, the sigma = 1 solve
                self.W = ctx.sparse_solve(desc, xh, xu, np.ascontiguousarray(y_cov_factor, dtype=np.float64), 0.0,
                                          1.0, jitter)
                self._state_variables |= {"W"}


class _LandmarksConditionalCholesky:
    _center_name = "landmarks"

    def _leverage(self, Xnew, sigma):       # conditional.py:908-922 (scalar sigma)
        return _landmarks_leverage(Xnew, self.landmarks, self.cov_func, _scalar_sigma(sigma), self.jitter,
                                   L=getattr(self, "L", None))

    def __init__(self, xu, pre_transformation, mu, cov_func, n_obs, L=None, sigma=DEFAULT_SIGMA,
                 jitter=DEFAULT_JITTER, y_is_mean=False, with_uncertainty=False, obs_variance=False,
                 obs_x=None, obs_y=None):
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
        if obs_variance:          # conditional.py:842-851,870-897: HC3 residuals of (obs_x, obs_y), second landmark GP
            if obs_x is None or obs_y is None:
                raise ValueError("obs_x and obs_y are required when obs_variance=True "
                                 "for LandmarksConditionalCholesky.")
            s = _scalar_sigma(sigma)
            if s is None or not s > 0:
                raise ValueError("obs_variance needs a positive noise level `sigma` "
                                 "(the reference divides by sigma^2, conditional.py:157-159).")
            ctx = _lib.default_context()
            ox = np.ascontiguousarray(ensure_2d(obs_x), dtype=np.float64)
            oy = np.asarray(obs_y, dtype=np.float64)
            h = self._leverage(ox, s)          # before `L` is attached: K_uu = cov(xu, xu), as in the reference
            self._corrected_r2 = _hc3(oy - self._mean(ox), h)
            self.variance_mu = 0.0
            self.variance_weights = ctx.sparse_solve(cov_func.lower(xu.shape[1]), ox, xu, self._corrected_r2,
                                                     self.variance_mu, s, jitter)
            self._state_variables |= {"variance_weights", "variance_mu"}
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
