"""FunctionEstimator (mellon/function_estimator.py): GP regression of y (n x p) on x with
noise sigma; no optimiser, only the conditional (function_estimator.py:318-374)."""
import logging

import numpy as np

from .base_model import BaseEstimator, DEFAULT_COV_FUNC
from .inference import DEFAULT_INIT_LEARN_RATE, DEFAULT_N_ITER, DEFAULT_OPTIMIZER, compute_conditional
from .parameters import DEFAULT_RANDOM_SEED
from .util import DEFAULT_JITTER, GaussianProcessType
from .validation import validate_array, validate_bool, validate_float, validate_float_or_iterable_numerical

logger = logging.getLogger("mellon")


class FunctionEstimator(BaseEstimator):
    """reference function_estimator.py:28-615."""

    def __init__(self, cov_func_curry=DEFAULT_COV_FUNC, n_landmarks=None, gp_type=None, jitter=DEFAULT_JITTER,
                 optimizer=DEFAULT_OPTIMIZER, n_iter=DEFAULT_N_ITER, init_learn_rate=DEFAULT_INIT_LEARN_RATE,
                 landmarks=None, nn_distances=None, mu=0, ls=None, ls_factor=1, cov_func=None, sigma=0,
                 y_is_mean=False, predictor_with_uncertainty=False, obs_variance=False, jit=True,
                 random_state=DEFAULT_RANDOM_SEED):
        super().__init__(cov_func_curry=cov_func_curry, n_landmarks=n_landmarks, rank=1.0, jitter=jitter,
                         gp_type=gp_type, landmarks=landmarks, nn_distances=nn_distances, mu=mu, ls=ls,
                         ls_factor=ls_factor, cov_func=cov_func,
                         predictor_with_uncertainty=predictor_with_uncertainty, jit=jit, random_state=random_state)
        self.y_is_mean = validate_bool(y_is_mean, "y_is_mean")
        self.mu = validate_float(mu, "mu")
        self.sigma = validate_float_or_iterable_numerical(sigma, "sigma", positive=True)
        self.obs_variance = validate_bool(obs_variance, "obs_variance")
        if self.gp_type in (GaussianProcessType.FULL_NYSTROEM, GaussianProcessType.SPARSE_NYSTROEM):
            raise ValueError(f"gp_type={gp_type} but the Nyström rank reduction is not available for the "
                             "Function Estimator. Use gp_type='cholesky' or gp_type='full' instead.")
        self.conditional = None
        self.y = None

    def __call__(self, x=None, y=None):
        return self.fit_predict(x=x, y=y)

    def prepare_inference(self, x):
        """reference function_estimator.py:295-316."""
        self.set_x(x)
        from .util import log_nn_new_fit
        log_nn_new_fit()
        try:
            self._prepare_attribute("n_landmarks")
            self._prepare_attribute("gp_type")
            if self.ls is None and self.cov_func is None:
                self._prepare_attribute("nn_distances")
            self._prepare_attribute("ls")
            self._prepare_attribute("cov_func")
            self._prepare_attribute("landmarks")
        finally:
            self._release_x_on_device()      # the HBM copy the 1-NN search / k-means shared: the conditional never uses it

    def _compute_ls(self):
        if self.cov_func is not None:
            return getattr(self.cov_func, "ls", 1.0)
        return super()._compute_ls()

    def compute_conditional(self, x=None, y=None, obs_variance=None):
        """reference function_estimator.py:318-374."""
        x = self.x if x is None else validate_array(x, "x")
        if x is None:
            raise ValueError("Required argument x is missing and self.x has not been set.")
        if y is None:
            raise ValueError("Required argument y is missing.")
        obs_variance = self.obs_variance if obs_variance is None else obs_variance
        self.conditional = compute_conditional(
            x, self.landmarks, None, None, y, self.mu, self.cov_func, None, None, self.sigma, jitter=self.jitter,
            y_is_mean=self.y_is_mean, with_uncertainty=self.predictor_with_uncertainty, obs_variance=obs_variance)
        return self.conditional

    def fit(self, x=None, y=None, obs_variance=None):
        x = self.set_x(x)
        y = validate_array(y, "y")
        if y.shape[0] != x.shape[0]:
            raise ValueError(f"X.shape[0] = {x.shape[0]:,} (n_samples) should equal y.shape[0] = {y.shape[0]:,}.")
        self.prepare_inference(x)
        self.compute_conditional(x, y, obs_variance=obs_variance)
        self.y = y
        return self

    @property
    def predict(self):
        if self.conditional is None:
            raise ValueError("The estimator has not been fitted: call fit(x, y) first.")
        return self.conditional

    def leverage(self, X=None):
        """Leverage with the fitted sigma, at the training cells by default (function_estimator.py:443-459)."""
        return self.predict.leverage(self.x if X is None else X)

    def loo_residuals_squared(self, X=None, y=None):
        """Squared leave-one-out residuals r^2 / (1 - h)^2; without arguments the values cached by an
        `obs_variance=True` fit (function_estimator.py:461-487)."""
        if X is None and y is None:
            if hasattr(self.predict, "_corrected_r2"):
                return self.predict._corrected_r2
            X, y = self.x, self.y
        else:
            X = self.x if X is None else X
            y = self.y if y is None else y
        return self.predict.loo_residuals_squared(X, y)

    def get_obs_variance(self, X=None):
        """Smoothed observation variance of the fitted predictor (function_estimator.py:489-505)."""
        return self.predict.obs_variance(self.x if X is None else X)

    def fit_predict(self, x=None, y=None, Xnew=None):
        self.fit(x, y)
        return self.predict(self.x if Xnew is None else validate_array(Xnew, "Xnew"))

    def multi_fit_predict(self, x=None, Y=None, Xnew=None):
        """Functions stored as ROWS of Y (reference function_estimator.py multi_fit_predict)."""
        Y = validate_array(Y, "Y")
        return self.fit_predict(x, Y.T, Xnew).T
