"""Inference (mellon/inference.py): transform, nearest-neighbour loss, MAP solve, diagonal
Laplace, log-density at the training cells and the conditional dispatch.  The loss, its gradient
and the Hessian diagonal are one fused device pass over L per evaluation (csrc/objective.hip);
the optimiser is SciPy's L-BFGS-B -- the routine the reference reaches through
jaxopt.ScipyMinimize (inference.py:285) -- driving that device objective from the host.
"""
import logging
from collections import namedtuple

import numpy as np
from scipy.optimize import minimize as _sp_minimize
from scipy.special import gammaln

from . import _lib
from .conditional import (ExpFullConditional, ExpLandmarksConditional, ExpLandmarksConditionalCholesky,
                          FullConditional, FullConditionalTime, LandmarksConditional,
                          LandmarksConditionalCholesky, LandmarksConditionalCholeskyTime,
                          LandmarksConditionalTime)
from .decomposition import FactorL, FactorLp
from .util import DEFAULT_JITTER, ensure_2d

logger = logging.getLogger("mellon")

DEFAULT_N_ITER = 100
DEFAULT_INIT_LEARN_RATE = 1e-1
DEFAULT_NUM_SAMPLES = 40
DEFAULT_OPTIMIZER = "L-BFGS-B"
DEFAULT_JIT = False

# The MAP objective is strictly convex (Hessian = I + L^T diag(e^{f+V}) L), so the optimum is
# unique.  The reference stops L-BFGS-B at SciPy's defaults (ftol 2.2e-9, gtol 1e-5, maxcor 10,
# maxiter 500), which leaves the log-density ~5e-5 (relative) short of that optimum and makes its
# output irreproducible below ~1e-4 across BLAS roundings (tests/test_oracle.py).  To honour
# "log-density within 1e-5" we converge to the optimum instead: SciPy's memory of 10 pairs (on the
# preconditioned variable 7 ... 30 pairs all need 46-49 passes at C3, tools/maxcor_sweep.py) and a tighter
# stopping rule reach ~3e-7 in far fewer evaluations than the reference's defaults use.
LBFGSB_OPTIONS = dict(maxiter=5000, maxfun=50000, maxcor=10, ftol=1e-13, gtol=1e-7)
REFERENCE_LBFGSB_OPTIONS = dict(maxiter=500)     # jaxopt.ScipyMinimize defaults


def _fit_of(L):
    if isinstance(L, (FactorL, FactorLp)):
        return L.fit
    return _lib.Fit.from_L(_lib.default_context(), np.asarray(L, dtype=np.float64))


class Transform:
    """z -> L z + mu (inference.py:51-69,125-139)."""

    def __init__(self, mu, L):
        self.mu = float(mu)
        self.L = L
        self.fit = _fit_of(L)

    def __call__(self, z):
        return self.fit.transform(np.asarray(z, dtype=np.float64), self.mu)


def compute_transform(mu, L):
    return Transform(mu, L)


def nn_likelihood_constants(nn_distances, d):
    """V and Vdr of inference.py:83-85 (d may be a per-cell vector)."""
    r = np.asarray(nn_distances, dtype=np.float64)
    d = np.asarray(d, dtype=np.float64)
    const = (d * np.log(np.pi) / 2) - gammaln(d / 2 + 1)
    from .util import log_nn
    log_r = log_nn(nn_distances)
    V = log_r * d + const
    Vdr = np.log(d) + ((d - 1) * log_r) + const
    return np.ascontiguousarray(np.broadcast_to(V, r.shape)), np.ascontiguousarray(np.broadcast_to(Vdr, r.shape))


class LossFunc:
    """loss(z) = -(prior(z) + likelihood(transform(z))) (inference.py:167-192).  Calling it
    returns the loss; `value_and_grad(z)` returns (loss, grad) from the same single device pass."""

    def __init__(self, nn_distances, d, transform, k, constants=None):
        self.fit = transform.fit
        self.k = int(k)
        if self.fit.m != self.k:
            raise ValueError(f"initial value has {k} entries but L has {self.fit.m} columns")
        V, Vdr = constants if constants is not None else nn_likelihood_constants(nn_distances, d)
        if V.shape[0] != self.fit.n:
            raise ValueError(f"{V.shape[0]} nearest-neighbour distances for {self.fit.n} rows of L")
        self.fit.set_likelihood(V, Vdr, transform.mu)
        self.n_eval = 0
        self.preconditioned = True     # optimise u with z = C^-T u, C C^T ~ L^T L + I (see minimize_lbfgsb)
        # L-BFGS inside the library; False (or more than 8192 landmarks, beyond the device-resident solver's register
        # budget): SciPy L-BFGS-B drives the device objective, one evaluation per call
        self.native_solver = self.fit.m <= 8192
        from .parameters import ridge_row_stride
        self.fit.precond_build(*ridge_row_stride(self.fit.n, self.fit.m, with_offset=True))   # no-op if the Ridge init built it

    def value_and_grad(self, z):
        self.n_eval += 1
        return self.fit.objective(z)

    # preconditioned variable (include/mellon_hip.h: mln_objective_precond)
    def value_and_grad_u(self, u):
        self.n_eval += 1
        loss, grad_u, self._last_z = self.fit.objective_precond(u)
        return loss, grad_u

    def u_from_z(self, z):
        return self.fit.precond_apply(0, z)

    def z_from_u(self, u):
        return self.fit.precond_apply(1, u)

    def hessian_diagonal(self, z):
        if getattr(self.fit, "implicit", False):
            raise NotImplementedError("the Hessian diagonal needs the explicit factor L "
                                      "(fit prepared with implicit=False; predictor_with_uncertainty does this)")
        return self.fit.objective(z, with_hess=True)[2]

    def __call__(self, z):
        return self.value_and_grad(z)[0]


def compute_loss_func(nn_distances, d, transform, k, constants=None):
    return LossFunc(nn_distances, d, transform, k, constants)


def minimize_lbfgsb(loss_func, initial_value, jit=DEFAULT_JIT, options=None):
    """inference.py:272-288.  `options` overrides LBFGSB_OPTIONS.  options="reference" runs the reference AS RUN:
    SciPy's L-BFGS-B (the routine behind jaxopt.ScipyMinimize) with its default stopping rule (ftol 2.2e-9, gtol 1e-5,
    maxcor 10, maxiter 500) on the un-preconditioned variable z, every evaluation one device pass -- the early-stopped
    answer of the reference (~5e-5 from the optimum) instead of the optimum."""
    reference_mode = isinstance(options, str) and options == "reference"
    opts = dict(LBFGSB_OPTIONS)
    if reference_mode:
        opts = dict(REFERENCE_LBFGSB_OPTIONS)
    elif options:
        opts.update(options)
    Results = namedtuple("Results", "pre_transformation opt_state loss")
    z0 = np.asarray(initial_value, dtype=np.float64)
    if getattr(loss_func, "preconditioned", False) and not reference_mode:
        # Same method, same objective, better-conditioned variable: z = C^-T u with
        # C C^T ~ L^T L + I (the Ridge matrix = the MAP Hessian where e^{f+V} = 1).  The optimum is
        # unique (strict convexity), so this only changes how many passes over L it takes (~10x fewer).
        if loss_func.native_solver:
            # L-BFGS inside libmellon_hip.so (mln_map_solve): one device pass per evaluation, no
            # Python / SciPy between evaluations
            z, loss, n_eval, n_iter, status = loss_func.fit.map_solve(
                z0, maxiter=opts["maxiter"], maxcor=opts["maxcor"], ftol=opts["ftol"], gtol=opts["gtol"])
            loss_func.n_eval += n_eval
            State = namedtuple("State", "fun_val nfev nit status success")
            capped, status = bool(status & 4), status & 3     # bit 2: stopped with cells above the likelihood cap (mln_map_solve)
            if status != 0 or not np.isfinite(loss):
                # (the reference's jaxopt wrapper reports non-convergence in its state only; a density that is not the MAP
                #  estimate deserves a line in the log)
                logger.warning("L-BFGS did not converge (status %d: %s) after %d evaluations, loss %.6g; the returned "
                               "pre_transformation is the last accepted point.", status,
                               {1: "iteration limit", 2: "line search failed"}.get(status, "?"), n_eval, loss)
                if capped:
                    logger.warning("The solve stopped with cells above the likelihood cap: the reported loss is the capped "
                                   "(quadratically continued) objective's, a lower bound of the reference's loss at that point.")
            return Results(z, State(loss, n_eval, n_iter, status, status == 0), float(loss))
        res = _sp_minimize(loss_func.value_and_grad_u, loss_func.u_from_z(z0), jac=True, method="L-BFGS-B",
                           options=opts)
        return Results(loss_func.z_from_u(res.x), res, float(res.fun))
    if hasattr(loss_func, "value_and_grad"):
        fun, jac = loss_func.value_and_grad, True
    else:
        fun, jac = loss_func, None                       # user callable: finite differences by SciPy
    res = _sp_minimize(fun, z0, jac=jac, method="L-BFGS-B", options=opts)
    return Results(res.x, res, float(res.fun))


def _unavailable_optimizer(name):
    def f(*a, **k):
        raise NotImplementedError(f"optimizer '{name}' is outside the accelerated path; use 'L-BFGS-B' "
                                  "(the MAP optimum is unique, SURVEY.md S8a-7).")
    return f


def minimize_adam(loss_func, initial_value, n_iter=DEFAULT_N_ITER, init_learn_rate=DEFAULT_INIT_LEARN_RATE,
                  jit=DEFAULT_JIT):
    """inference.py:222-269: `n_iter` Adam steps (jax.example_libraries.optimizers.adam: b1 = 0.9, b2 = 0.999,
    eps = 1e-8, bias-corrected moments) with the learning rate exp(-0.01 i) * init_learn_rate; every step is one
    fused loss-and-gradient pass on the device.  Returns (pre_transformation, opt_state, losses)."""
    z = np.array(initial_value, dtype=np.float64)
    m1, m2 = np.zeros_like(z), np.zeros_like(z)
    b1, b2, eps = 0.9, 0.999, 1e-8
    losses = []
    fun = loss_func.value_and_grad if hasattr(loss_func, "value_and_grad") else None
    if fun is None:
        raise TypeError("minimize_adam needs a loss with value_and_grad (compute_loss_func returns one)")
    for i in range(int(n_iter)):
        value, g = fun(z)
        losses.append(float(value))
        m1 = (1 - b1) * g + b1 * m1
        m2 = (1 - b2) * np.square(g) + b2 * m2
        mhat = m1 / (1 - b1 ** (i + 1))
        vhat = m2 / (1 - b2 ** (i + 1))
        z = z - np.exp(-1e-2 * i) * init_learn_rate * mhat / (np.sqrt(vhat) + eps)
    Results = namedtuple("Results", "pre_transformation opt_state losses")
    return Results(z, (z, m1, m2), np.asarray(losses))


run_advi = _unavailable_optimizer("advi")


def compute_laplace_std(loss_func, pre_transformation, jit=DEFAULT_JIT):
    """inference.py:291-338: 1 / sqrt(max(diag H, 1e-8)); diag H in closed form from one pass over L."""
    h = np.maximum(loss_func.hessian_diagonal(np.asarray(pre_transformation, dtype=np.float64)), 1e-8)
    stds = 1.0 / np.sqrt(h)
    logger.info("Laplace approximation: Hessian diagonal range [%.3e, %.3e], std range [%.3e, %.3e].",
                float(h.min()), float(h.max()), float(stds.min()), float(stds.max()))
    return stds


def compute_log_density_x(pre_transformation, transform):
    """inference.py:341-354."""
    return transform(pre_transformation)


def _dispatch(classes, x, landmarks, pre_transformation, pre_transformation_std, y, mu, cov_func, L, Lp, sigma,
              jitter, y_is_mean, with_uncertainty, obs_variance):
    Full, Landmarks, Cholesky = classes
    if landmarks is None:
        logger.debug("Using FullConditional GP.")
        return Full(x, y, mu, cov_func, Lp, sigma=sigma, jitter=jitter, y_is_mean=y_is_mean,
                    with_uncertainty=with_uncertainty and (pre_transformation_std is not None or not y_is_mean),
                    obs_variance=obs_variance, parameter_std=pre_transformation_std, factor=L)
    landmarks = ensure_2d(landmarks)
    if pre_transformation is not None and np.shape(pre_transformation)[0] == landmarks.shape[0]:
        logger.debug("Using LandmarksConditionalCholesky GP.")
        if pre_transformation_std is not None and sigma is not None and np.any(np.asarray(sigma) > 0):
            raise ValueError("One can specify either `sigma` or `pre_transformation_std` "
                             "to describe uncertainty, but not both.")
        if pre_transformation_std is not None:
            sigma = pre_transformation_std               # inference.py:470-471
        return Cholesky(landmarks, pre_transformation, mu, cov_func, x.shape[0], Lp, sigma=sigma, jitter=jitter,
                        y_is_mean=y_is_mean, with_uncertainty=with_uncertainty, obs_variance=obs_variance,
                        obs_x=x if obs_variance else None, obs_y=y if obs_variance else None)
    logger.debug("Using LandmarksConditional GP.")
    # inference.py:488-492: y_cov_factor = L diag(pre_transformation_std) -- formed where it is consumed
    return Landmarks(x, landmarks, y, mu, cov_func, L, sigma=sigma, jitter=jitter, y_is_mean=y_is_mean,
                     with_uncertainty=with_uncertainty, obs_variance=obs_variance,
                     parameter_std=pre_transformation_std if with_uncertainty else None)


def compute_conditional(x, landmarks, pre_transformation, pre_transformation_std, y, mu, cov_func, L, Lp=None,
                        sigma=0, jitter=DEFAULT_JITTER, y_is_mean=False, with_uncertainty=False,
                        obs_variance=False):
    """inference.py:375-508."""
    return _dispatch((FullConditional, LandmarksConditional, LandmarksConditionalCholesky), x, landmarks,
                     pre_transformation, pre_transformation_std, y, mu, cov_func, L, Lp, sigma, jitter, y_is_mean,
                     with_uncertainty, obs_variance)


def compute_conditional_times(x, landmarks, pre_transformation, pre_transformation_std, y, mu, cov_func, L,
                              Lp=None, sigma=0, jitter=DEFAULT_JITTER, y_is_mean=False, with_uncertainty=False,
                              obs_variance=False):
    """inference.py:511-640 (time-aware predictors)."""
    return _dispatch((FullConditionalTime, LandmarksConditionalTime, LandmarksConditionalCholeskyTime), x,
                     landmarks, pre_transformation, pre_transformation_std, y, mu, cov_func, L, Lp, sigma, jitter,
                     y_is_mean, with_uncertainty, obs_variance)


def compute_conditional_explog(x, landmarks, pre_transformation, pre_transformation_std, y, mu, cov_func, L,
                               Lp=None, sigma=0, jitter=DEFAULT_JITTER, y_is_mean=False, with_uncertainty=False,
                               obs_variance=False):
    """inference.py:643-765 (exp of the predicted log value)."""
    return _dispatch((ExpFullConditional, ExpLandmarksConditional, ExpLandmarksConditionalCholesky), x, landmarks,
                     pre_transformation, pre_transformation_std, y, mu, cov_func, L, Lp, sigma, jitter, y_is_mean,
                     with_uncertainty, obs_variance)
