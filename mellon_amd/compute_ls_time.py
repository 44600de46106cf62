"""Length scale of the time axis for TimeSensitiveDensityEstimator when `ls_time` is not given
(reference: mellon/compute_ls_time.py:12-105).

Idea of the reference: fit one density per time point, evaluate every one of them on ALL cells, and pick the kernel
length scale whose covariance over the time gaps |t_a - t_b| best matches (Frobenius norm) the correlation of those
log-densities.  Here the per-time-point fits and their cross-evaluation are device work (each fit is the hot path on
that time point's cells; each evaluation one fused predict pass), the T x T correlation and the one-dimensional
search are host arithmetic on a handful of numbers.
"""
import logging

import numpy as np
from scipy.optimize import minimize

from .validation import validate_time_x

logger = logging.getLogger("mellon")

SMALL_TIME_POINT = 500      # cells below which the reference warns about an unreliable time point


def _fit_time_points(states, stamp, nn_distances, estimator_kwargs):
    """One DensityEstimator per distinct time stamp, fitted on that time point's cells only.
    Returns (sorted unique stamps, fitted estimators)."""
    from .density_estimator import DensityEstimator
    stamps = np.unique(stamp)
    fitted = []
    for rank, t in enumerate(stamps, start=1):
        members = np.flatnonzero(stamp == t)
        logger.info(f"[{rank} of {len(stamps)}] Computing density for {members.size:,} cells at time point {t}.")
        if members.size < SMALL_TIME_POINT:
            logger.warning(f"Time point {t} only has {members.size:,} cells. "
                           "This could lead to inaccurate estimation of the time length scale `ls_time`.")
        model = DensityEstimator(nn_distances=nn_distances[members], **estimator_kwargs)
        model.fit(np.ascontiguousarray(states[members]))
        fitted.append(model)
    return stamps, fitted


def _correlation_of_rows(table):
    """Pearson correlation between the rows of `table` (T x n): centred rows, normalised Gram."""
    centred = table - table.mean(axis=1, keepdims=True)
    gram = centred @ centred.T
    scale = np.sqrt(np.diag(gram))
    return gram / np.outer(scale, scale)


def _closest_length_scale(cov_func_curry, stamps, target):
    """argmin over ls of || k_ls(|t_a - t_b|) - target ||_F, searched in log ls from ls = 1 with L-BFGS-B (the
    reference's jaxopt.ScipyMinimize(method="L-BFGS-B").run(0.0); the derivative comes from differences here).
    The kernel is evaluated on the DISTINCT gaps only and scattered into the T x T matrix."""
    gaps = np.abs(stamps[:, None] - stamps[None, :])
    distinct, where = np.unique(gaps, return_inverse=True)
    where = where.reshape(gaps.shape)
    column, origin = distinct.reshape(-1, 1), np.zeros((1, 1))

    def misfit(log_ls):
        kernel = cov_func_curry(float(np.exp(np.ravel(log_ls)[0])))
        values = np.asarray(kernel(column, origin)).reshape(-1)
        return float(np.linalg.norm(values[where] - target))

    found = minimize(misfit, np.zeros(1), method="L-BFGS-B")
    return float(np.exp(found.x[0]))


def compute_ls_time(nn_distances, x, cov_func_curry, times=None, warn_below=SMALL_TIME_POINT, return_data=False,
                    density_estimator_kwargs=None):
    """ls_time, or with `return_data` the tuple (ls_time, densities [T x n], per-time-point estimators, time points)."""
    global SMALL_TIME_POINT
    xt = np.asarray(validate_time_x(x, times), dtype=np.float64)
    states, stamp = np.ascontiguousarray(xt[:, :-1]), xt[:, -1]
    nn_distances = np.asarray(nn_distances, dtype=np.float64)
    previous, SMALL_TIME_POINT = SMALL_TIME_POINT, warn_below
    try:
        stamps, fitted = _fit_time_points(states, stamp, nn_distances, dict(density_estimator_kwargs or {}))
    finally:
        SMALL_TIME_POINT = previous
    # every time point's density evaluated on the cells of ALL time points (one fused device pass each)
    table = np.stack([model.predict(states) for model in fitted])
    ls_time = _closest_length_scale(cov_func_curry, stamps, _correlation_of_rows(table))
    if return_data:
        return ls_time, table, fitted, stamps
    return ls_time
