"""DensityEstimator (mellon/density_estimator.py): same constructor, attributes and
fit / fit_predict / predict / prepare_inference / run_inference / process_inference flow; the
covariance factorisation, the MAP objective and the predictor run on the MI355X."""
import logging

import numpy as np

from .base_model import BaseEstimator, DEFAULT_COV_FUNC
from .inference import (DEFAULT_INIT_LEARN_RATE, DEFAULT_JIT, DEFAULT_N_ITER, DEFAULT_OPTIMIZER,
                        compute_conditional, compute_log_density_x, compute_loss_func, compute_transform)
from .parameters import DEFAULT_RANDOM_SEED, compute_d, compute_initial_value, compute_mu
from .util import DEFAULT_JITTER
from .validation import validate_array, validate_float_or_iterable_numerical, validate_string

logger = logging.getLogger("mellon")


class _Background:
    """Run one call in a worker thread and re-raise its exception on join()."""

    def __init__(self, fn):
        import threading
        from . import distributed
        self._exc, self._fn = None, fn
        self._state = distributed.thread_state()      # the worker acts for the same rank / device context
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _run(self):
        try:
            from . import distributed
            distributed.adopt_thread_state(self._state)
            self._fn()
        except BaseException as e:     # noqa: BLE001 -- re-raised in join()
            self._exc = e

    def join(self):
        self._t.join()
        if self._exc is not None:
            exc, self._exc = self._exc, None
            raise exc


class DensityEstimator(BaseEstimator):
    """Non-parametric cell-state density estimator (reference density_estimator.py:35-581)."""

    def __init__(self, cov_func_curry=DEFAULT_COV_FUNC, n_landmarks=None, rank=None, gp_type=None, d_method=None,
                 jitter=DEFAULT_JITTER, optimizer=DEFAULT_OPTIMIZER, n_iter=DEFAULT_N_ITER,
                 init_learn_rate=DEFAULT_INIT_LEARN_RATE, landmarks=None, nn_distances=None, d=None, mu=None,
                 ls=None, ls_factor=1, cov_func=None, Lp=None, L=None, initial_value=None,
                 predictor_with_uncertainty=False, jit=DEFAULT_JIT, check_rank=None,
                 random_state=DEFAULT_RANDOM_SEED):
        super().__init__(cov_func_curry=cov_func_curry, n_landmarks=n_landmarks, rank=rank, jitter=jitter,
                         gp_type=gp_type, optimizer=optimizer, n_iter=n_iter, init_learn_rate=init_learn_rate,
                         landmarks=landmarks, nn_distances=nn_distances, d=d, mu=mu, ls=ls, ls_factor=ls_factor,
                         cov_func=cov_func, Lp=Lp, L=L, initial_value=initial_value,
                         predictor_with_uncertainty=predictor_with_uncertainty, jit=jit, check_rank=check_rank,
                         random_state=random_state)
        if d_method is None:
            d_method = "manual" if d is not None else "embedding"
        self.d_method = validate_string(d_method, "d_method", choices={"fractal", "embedding", "manual"})
        if self.d_method == "manual" and d is None:
            raise ValueError("d_method='manual' requires d.")
        self.transform = None
        self.loss_func = None
        self.opt_state = None
        self.losses = None
        self.log_density_x = None
        self.log_density_func = None

    # -- attribute computations (reference density_estimator.py:311-402) --------------------------------
    def _compute_d(self):
        if self.d_method == "fractal":
            raise NotImplementedError("d_method='fractal' (util.local_dimensionality) is outside the accelerated path.")
        d = self.d if self.d_method == "manual" else compute_d(self.x)
        logger.info(f"Using d={d}.")
        if np.ndim(d) == 0 and d > 50:
            raise ValueError(
                "The detected dimensionality of the data is over 50, which is likely to cause numerical "
                f"instability issues. Consider running a dimensionality reduction algorithm, or if this number "
                f"of dimensions is intended, explicitly pass d={self.d} as a parameter.")
        return d

    def _compute_mu(self):
        return compute_mu(self.nn_distances, self.d)

    def _compute_initial_value(self):
        # lbfgsb_options = "reference": the reference's exact Ridge on all cells is part of "as run"
        # ... and so is it for the optimisers that do not run to convergence (adam: a fixed number of steps), whose result
        # depends on where they start; the converged L-BFGS route may start anywhere (unique optimum) and takes the
        # sampled Gram of the preconditioner
        exact = (isinstance(self.lbfgsb_options, str) and self.lbfgsb_options == "reference") or \
            str(getattr(self, "optimizer", "L-BFGS-B")).lower() not in ("l-bfgs-b", "lbfgsb")
        return compute_initial_value(self.nn_distances, self.d, self.mu, self.L, row_stride=1 if exact else None,
                                     target=getattr(self, "_ridge_target", None))

    def _compute_transform(self):
        return compute_transform(self.mu, self.L)

    def _compute_loss_func(self):
        return compute_loss_func(self.nn_distances, self.d, self.transform, self.initial_value.shape[0],
                                 constants=getattr(self, "_lik_constants", None))

    def _set_log_density_x(self):
        self.log_density_x = compute_log_density_x(self.pre_transformation, self.transform)

    def _build_conditional(self):
        return compute_conditional(self.x, self.landmarks, self.pre_transformation, self.pre_transformation_std,
                                   self.log_density_x, self.mu, self.cov_func, self.L, self.Lp, sigma=None,
                                   jitter=self.jitter, y_is_mean=True,
                                   with_uncertainty=self.predictor_with_uncertainty)

    def _set_log_density_func(self):
        logger.info("Computing predictive function.")
        f = self._build_conditional()
        f.n_obs = self._n_obs()
        f.d = self.d
        f.d_method = self.d_method
        self.log_density_func = f

    def _n_obs(self):
        return self.x.shape[0]

    # -- public flow (reference density_estimator.py:404-581) ---------------------------------------------
    _PIPELINE = ("n_landmarks", "rank", "gp_type", None, "nn_distances", "d", "mu", "ls", "cov_func", "landmarks",
                 "Lp", "L", "initial_value", "transform", "loss_func")

    def _validate_x_arg(self, x):
        return validate_array(x, "x")

    def prepare_inference(self, x):
        if x is None:
            if self.x is None:
                raise ValueError("Required argument x is missing and self.x has not been set.")
            x = self.x
        elif self.x is not None and self.x is not x:
            raise ValueError("self.x has been set already, but is not equal to the argument x.")
        self.set_x(x)
        from .util import log_nn_new_fit
        log_nn_new_fit()                     # (the host prologue's cached logarithms belong to ONE fit)
        try:
            return self._prepare_pipeline()
        finally:
            self._release_x_on_device()      # the one HBM copy of host cells the steps before the fit shared

    def _prepare_pipeline(self):
        worker = None
        for attr in self._PIPELINE:
            if attr is None:
                self.validate_parameter()
            elif attr == "mu" and worker is None:
                # The device factorisation (Lp, K) only needs the covariance and the landmarks; the
                # O(n) host heuristics behind mu / the likelihood constants do not depend on it.
                # Run the former in a worker thread (ctypes drops the GIL) while the host does the latter.
                # (With ranks the heuristics' reductions travel over the host communicator, never through the
                #  device context the worker is using.)
                for early in self._DEVICE_FIT_INPUTS:
                    self._prepare_attribute(early)
                worker = _Background(self._device_fit_and_preconditioner)
                try:
                    self._prepare_attribute("mu")
                    self._host_constants()
                except BaseException:
                    try:
                        worker.join()                  # never leave device work running on the shared context
                    except BaseException:              # noqa: BLE001 -- the primary error wins
                        pass
                    raise
            elif attr == "Lp" and worker is not None:
                worker.join()
                worker = None
                self._prepare_attribute(attr)
            else:
                self._prepare_attribute(attr)
        if worker is not None:
            worker.join()
        return self.loss_func, self.initial_value

    _DEVICE_FIT_INPUTS = ("ls", "cov_func", "landmarks")

    def _device_fit_and_preconditioner(self):
        """The worker's share of the pipeline: the factorisation handle and -- on the default route, single rank -- the first
        preconditioner too (its Gram and factorisation chain need the covariance and the landmarks only; `mu`, which the
        Ridge start waits for, is being computed by the calling thread meanwhile).  On small problems the host heuristics
        outlast the kernel-matrix pass and the device would sit idle until they are done (C2: 0.4 ms of an 8 ms step).
        compute_initial_value() finds the factor in place (precond_build is a no-op for the stride it already has)."""
        fit = self._device_fit()
        from .distributed import current
        from .parameters import ridge_row_stride
        exact = (isinstance(self.lbfgsb_options, str) and self.lbfgsb_options == "reference") or \
            str(getattr(self, "optimizer", "L-BFGS-B")).lower() not in ("l-bfgs-b", "lbfgsb")
        if current().world_size == 1 and not exact and self.initial_value is None and self.L is None \
                and getattr(fit, "m", 0) and getattr(fit, "n", 0) > fit.m and hasattr(fit, "precond_build"):
            stride, offset = ridge_row_stride(fit.n, fit.m, with_offset=True)
            fit.precond_build(stride, offset)
        return fit

    def _host_constants(self):
        """mle - mu and the likelihood constants (inference.py:83-85), computed once."""
        from .inference import nn_likelihood_constants
        from .util import mle
        if self.nn_distances is not None and self.d is not None and self.mu is not None:
            self._lik_constants = nn_likelihood_constants(self.nn_distances, self.d)
            self._ridge_target = -self._lik_constants[0] - self.mu        # mle = -V  (util.py:348)
        return None

    def run_inference(self, loss_func=None, initial_value=None, optimizer=None):
        if loss_func is not None:
            self.loss_func = loss_func
        if initial_value is not None:
            self.initial_value = initial_value
        if optimizer is not None:
            switched = str(optimizer).lower() != str(self.optimizer).lower()
            self.optimizer = optimizer
            # An optimiser that does not run to convergence (adam: a fixed number of steps) ends where its start puts
            # it: it gets the reference's exact Ridge start (all cells), also when prepare_inference() ran under another
            # optimiser and built the start from the sampled Gram.
            # (only a start the estimator derived itself: one the user passed or assigned is kept, as in the reference)
            if switched and initial_value is None and str(optimizer).lower() not in ("l-bfgs-b", "lbfgsb") \
                    and self._is_derived("initial_value") \
                    and self.L is not None and self.nn_distances is not None:
                self.initial_value = None
                self._prepare_attribute("initial_value")
        self._run_inference()
        return self.pre_transformation

    def process_inference(self, pre_transformation=None, build_predict=True):
        if pre_transformation is not None:
            self.pre_transformation = validate_array(pre_transformation, "pre_transformation")
        self._set_log_density_x()
        if build_predict:
            self._set_log_density_func()
        return self.log_density_x

    def fit(self, x=None, build_predict=True):
        self.prepare_inference(x)
        self.run_inference()
        self.process_inference(build_predict=build_predict)
        return self

    @property
    def predict(self):
        if self.log_density_func is None:
            self._set_log_density_func()
        return self.log_density_func

    def fit_predict(self, x=None, build_predict=False):
        if self.x is not None and x is not None and self.x is not x:
            raise ValueError("self.x has been set already, but is not equal to the argument x.")
        if self.x is None and x is None:
            raise ValueError("Required argument x is missing and self.x has not been set.")
        if x is None:
            x = self.x
        else:
            x = self._validate_x_arg(x)
        self.fit(x, build_predict=build_predict)
        return self.log_density_x
