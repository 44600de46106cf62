"""TimeSensitiveDensityEstimator (mellon/time_sensitive_density_estimator.py): the density path
with a time column, per-time-point nearest-neighbour distances and the product kernel
cov(ls, active_dims=:-1) * cov(ls_time, active_dims=-1) (parameters.py:641-644)."""
import logging

import numpy as np

from .base_model import DEFAULT_COV_FUNC
from .density_estimator import DensityEstimator
from .inference import (DEFAULT_INIT_LEARN_RATE, DEFAULT_JIT, DEFAULT_N_ITER, DEFAULT_OPTIMIZER,
                        compute_conditional_times)
from .parameters import (DEFAULT_RANDOM_SEED, compute_average_cell_count, compute_cov_func, compute_d, compute_ls,
                         compute_landmarks_rescale_time, compute_nn_distances_within_time_points)
from .util import DEFAULT_JITTER
from .validation import validate_nn_distances_sharded, validate_positive_float, validate_time_x

logger = logging.getLogger("mellon")


class TimeSensitiveDensityEstimator(DensityEstimator):
    """reference time_sensitive_density_estimator.py:39-796 (fit / predict on [x | time])."""

    def __init__(self, cov_func_curry=DEFAULT_COV_FUNC, n_landmarks=None, rank=None, gp_type=None, d_method=None,
                 jitter=DEFAULT_JITTER, optimizer=DEFAULT_OPTIMIZER, n_iter=DEFAULT_N_ITER,
                 init_learn_rate=DEFAULT_INIT_LEARN_RATE, landmarks=None, nn_distances=None,
                 normalize_per_time_point=False, d=None, mu=None, ls=None, ls_factor=1, ls_time=None,
                 ls_time_factor=1, density_estimator_kwargs=None, cov_func=None, Lp=None, L=None, initial_value=None,
                 predictor_with_uncertainty=False, jit=DEFAULT_JIT, check_rank=None,
                 random_state=DEFAULT_RANDOM_SEED, _save_intermediate_ls_times=False):
        super().__init__(cov_func_curry=cov_func_curry, n_landmarks=n_landmarks, rank=rank, gp_type=gp_type,
                         d_method=d_method, jitter=jitter, optimizer=optimizer, n_iter=n_iter,
                         init_learn_rate=init_learn_rate, landmarks=landmarks, nn_distances=nn_distances, d=d,
                         mu=mu, ls=ls, ls_factor=ls_factor, cov_func=cov_func, Lp=Lp, L=L,
                         initial_value=initial_value, predictor_with_uncertainty=predictor_with_uncertainty,
                         jit=jit, check_rank=check_rank, random_state=random_state)
        self.normalize_per_time_point = normalize_per_time_point
        self.ls_time = validate_positive_float(ls_time, "ls_time", optional=True)
        self.ls_time_factor = validate_positive_float(ls_time_factor, "ls_time_factor")
        if density_estimator_kwargs is None:
            density_estimator_kwargs = {}
        if not isinstance(density_estimator_kwargs, dict):
            raise ValueError("density_estimator_kwargs needs to be a dictionary.")
        self.density_estimator_kwargs = density_estimator_kwargs
        self._save_intermediate_ls_times = _save_intermediate_ls_times

    _PIPELINE = ("n_landmarks", "rank", "gp_type", None, "d", "nn_distances", "mu", "ls", "ls_time", "cov_func",
                 "landmarks", "Lp", "L", "initial_value", "transform", "loss_func")

    # mu before ls_time: the per-time-point fits behind the automatic ls_time receive the GLOBAL mu
    # (time_sensitive_density_estimator.py:655-657 prepares mu before ls / ls_time)
    _DEVICE_FIT_INPUTS = ("ls", "mu", "ls_time", "cov_func", "landmarks")

    def _compute_d(self):
        if self.d_method == "fractal":
            raise NotImplementedError("d_method='fractal' is outside the accelerated path.")
        d = self.d if self.d_method == "manual" else compute_d(self.x[:, :-1])
        if np.ndim(d) == 0 and d > 50:
            raise ValueError("The detected dimensionality of the data is over 50, which is likely to cause "
                             f"numerical instability issues; explicitly pass d={self.d} if intended.")
        return d

    def _nn_within_time_points(self, normalize):
        from .distributed import current
        if current().world_size == 1:
            return compute_nn_distances_within_time_points(self.x, d=self.d, normalize=normalize)
        x_all, lo = self._all_cells()                 # every time point needs its cells from all ranks
        return compute_nn_distances_within_time_points(x_all, d=self.d, normalize=normalize,
                                                       local=(lo, self.x.shape[0]))

    def _compute_nn_distances(self):
        logger.info("Computing nearest neighbor distances within time points.")
        from .distributed import current
        return validate_nn_distances_sharded(self._nn_within_time_points(self.normalize_per_time_point), current())

    def _compute_ls(self):
        nn = self.nn_distances
        if self.normalize_per_time_point is not False and self.normalize_per_time_point is not None:
            nn = self._nn_within_time_points(False)
        return compute_ls(nn) * self.ls_factor

    def _compute_ls_time(self):
        """time_sensitive_density_estimator.py:503-536: one density fit per time point, then the kernel length scale
        that best explains the correlation of the densities across time."""
        from .compute_ls_time import compute_ls_time
        # The per-time-point fits run on whatever cells this rank holds: with sharded cells every rank would loop over
        # its own set of time stamps (mismatched collectives) and correlate only local cells (a different ls_time, hence
        # a different cov_func, per rank).  The device context's communicator is bound to all ranks, so a rank-0-only
        # nested fit is not available either: the caller passes ls_time (a scalar, replicated) for sharded fits.
        self._require_single_process("ls_time")
        kwargs = {"cov_func_curry": self.cov_func_curry, "d_method": self.d_method, "d": self.d,
                  "optimizer": self.optimizer, "ls": self.ls, "ls_factor": self.ls_factor, "jit": self.jit,
                  "mu": self.mu}
        kwargs.update(self.density_estimator_kwargs)
        logger.info("Initiating density computation for each time point to estimate the 'ls_time' parameter. "
                    "You can directly specify 'ls_time' to bypass this computation-intensive step.")
        ls = compute_ls_time(self.nn_distances, self.x, self.cov_func_curry,
                             return_data=self._save_intermediate_ls_times, density_estimator_kwargs=kwargs)
        if self._save_intermediate_ls_times:
            logger.info("Storing `self.densities`, `self.predictors`, and `self.numeric_stages`.")
            ls, self.densities, self.predictors, self.numeric_stages = ls
        return ls * self.ls_time_factor

    def _compute_landmarks(self):
        from .distributed import current
        comm = current()
        if comm.world_size == 1:
            return compute_landmarks_rescale_time(self.x, self.ls, self.ls_time, n_landmarks=self.n_landmarks,
                                                  random_state=self._seed())
        x_all, _ = self._all_cells()
        lm = compute_landmarks_rescale_time(x_all, self.ls, self.ls_time, n_landmarks=self.n_landmarks,
                                            random_state=self._seed()) if comm.rank == 0 else None
        return comm.broadcast(None if lm is None else np.ascontiguousarray(lm, dtype=np.float64), src=0)

    def _compute_cov_func(self):
        cov_func = compute_cov_func(self.cov_func_curry, self.ls, self.ls_time)
        logger.info("Using covariance function %s.", str(cov_func))
        return cov_func

    def _build_conditional(self):
        return compute_conditional_times(self.x, self.landmarks, self.pre_transformation,
                                         self.pre_transformation_std, self.log_density_x, self.mu, self.cov_func,
                                         self.L, self.Lp, sigma=None, jitter=self.jitter, y_is_mean=True,
                                         with_uncertainty=self.predictor_with_uncertainty)

    def _n_obs(self):
        return compute_average_cell_count(self.x, self.normalize_per_time_point)

    def prepare_inference(self, x, times=None):
        if x is not None:
            xt = validate_time_x(x, times)
            if self.x is not None and self.x is not xt and not (times is None and self.x is x):
                raise ValueError("self.x has been set already, but is not equal to the argument x.")
            x = xt if self.x is None else self.x
        return super().prepare_inference(x)

    def fit(self, x=None, times=None, build_predict=True):
        self.prepare_inference(x, times)
        self.run_inference()
        self.process_inference(build_predict=build_predict)
        return self

    def fit_predict(self, x=None, times=None, build_predict=False):
        self.fit(x, times, build_predict=build_predict)
        return self.log_density_x
