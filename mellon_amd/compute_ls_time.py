"""Length scale of the time dimension (mellon/compute_ls_time.py:12-105): one density fit per time point
(each a device fit), the correlation of the predicted log-densities between time points, and the kernel
length scale whose covariance over the time gaps is closest to those correlations."""
import logging

import numpy as np
from scipy.optimize import minimize

from .validation import validate_time_x

logger = logging.getLogger("mellon")


def compute_ls_time(nn_distances, x, cov_func_curry, times=None, warn_below=500, return_data=False,
                    density_estimator_kwargs=None):
    """compute_ls_time.py:12-105.  Returns ls_time, and with `return_data` also (densities, predictors,
    unique_times)."""
    from .density_estimator import DensityEstimator

    x = np.asarray(validate_time_x(x, times), dtype=np.float64)
    nn_distances = np.asarray(nn_distances, dtype=np.float64)
    times_col, states = x[:, -1], np.ascontiguousarray(x[:, :-1])
    unique_times = np.unique(times_col)
    n_times = len(unique_times)
    densities, predictors = [], []
    for i, time in enumerate(unique_times):
        mask = times_col == time
        n_cells = int(mask.sum())
        logger.info(f"[{i + 1} of {n_times}] Computing density for {n_cells:,} cells at time point {time}.")
        if n_cells < warn_below:
            logger.warning(f"Time point {time} only has {n_cells:,} cells. "
                           "This could lead to inaccurate estimation of the time length scale `ls_time`.")
        est = DensityEstimator(nn_distances=nn_distances[mask], **(density_estimator_kwargs or {}))
        est.fit(np.ascontiguousarray(states[mask]))
        densities.append(est.predict(states))
        predictors.append(est)
    densities = np.stack(densities)
    corrs = np.corrcoef(densities)
    delta_t = np.abs(unique_times.reshape(-1, 1) - unique_times.reshape(1, -1)).reshape(-1, 1)
    origin = np.zeros((1, 1))

    def ls_loss(log_ls):
        ls = float(np.exp(np.ravel(log_ls)[0]))
        covs = np.asarray(cov_func_curry(ls)(delta_t, origin)).reshape((n_times, n_times))
        return float(np.linalg.norm(covs - corrs))

    # jaxopt.ScipyMinimize(method="L-BFGS-B").run(0.0) -- SciPy's defaults, gradient by differences instead of autodiff
    opt = minimize(ls_loss, np.array([0.0]), method="L-BFGS-B")
    ls = float(np.exp(opt.x[0]))
    if return_data:
        return ls, densities, predictors, unique_times
    return ls
