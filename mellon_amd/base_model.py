"""BaseEstimator (mellon/base_model.py): constructor validation, the lazy `_prepare_attribute`
pipeline and the optimiser switch.  Every attribute of the reference is kept and can be injected
through the constructor; `Lp` / `L` are device-resident factors (decomposition.FactorLp/FactorL)."""
import logging

import numpy as np

from . import _lib
from .cov import Matern52
from .decomposition import FactorL, FactorLp, _full_decomposition_low_rank, _modified_low_rank
from .inference import (DEFAULT_INIT_LEARN_RATE, DEFAULT_JIT, DEFAULT_N_ITER, DEFAULT_OPTIMIZER,
                        compute_laplace_std, minimize_adam, minimize_lbfgsb, run_advi)
from .parameter_validation import validate_cov_func, validate_cov_func_curry, validate_params
from .parameters import (DEFAULT_RANDOM_SEED, compute_cov_func, compute_gp_type, compute_landmarks,
                         compute_ls, compute_n_landmarks, compute_nn_distances, compute_rank, landmarks_backend)
from .util import DEFAULT_JITTER, GaussianProcessType, ensure_2d
from .validation import (validate_array, validate_bool, validate_float, validate_float_or_int,
                         validate_float_or_iterable_numerical, validate_nn_distances, validate_nn_distances_sharded,
                         validate_positive_float,
                         validate_positive_int, validate_string)

DEFAULT_COV_FUNC = Matern52
RANK_FRACTION_THRESHOLD = 0.8
SAMPLE_LANDMARK_RATIO = 10

logger = logging.getLogger("mellon")


class BaseEstimator:
    """Base class of the estimators (reference base_model.py:56-446)."""

    def __init__(self, cov_func_curry=DEFAULT_COV_FUNC, n_landmarks=None, rank=None, jitter=DEFAULT_JITTER,
                 optimizer=DEFAULT_OPTIMIZER, n_iter=DEFAULT_N_ITER, init_learn_rate=DEFAULT_INIT_LEARN_RATE,
                 landmarks=None, gp_type=None, nn_distances=None, d=None, mu=0, ls=None, ls_factor=1,
                 cov_func=None, Lp=None, L=None, initial_value=None, predictor_with_uncertainty=False,
                 jit=DEFAULT_JIT, check_rank=None, random_state=DEFAULT_RANDOM_SEED):
        self.cov_func_curry = validate_cov_func_curry(cov_func_curry, cov_func, "cov_func_curry")
        self.n_landmarks = validate_positive_int(n_landmarks, "n_landmarks", optional=True)
        self.random_state = validate_positive_int(random_state, "random_state", optional=True)
        self.rank = validate_float_or_int(rank, "rank", optional=True)
        self.jitter = validate_positive_float(jitter, "jitter")
        self.landmarks = validate_array(landmarks, "landmarks", optional=True)
        self.gp_type = GaussianProcessType.from_string(gp_type, optional=True)
        self.nn_distances = validate_nn_distances(validate_array(nn_distances, "nn_distances", optional=True),
                                                  optional=True)
        self.mu = validate_float(mu, "mu", optional=True)
        self.ls = validate_positive_float(ls, "ls", optional=True)
        self.ls_factor = validate_positive_float(ls_factor, "ls_factor")
        self.cov_func = validate_cov_func(cov_func, "cov_func", optional=True)
        self.Lp = Lp if isinstance(Lp, FactorLp) else validate_array(Lp, "Lp", optional=True)
        self.L = L if isinstance(L, (FactorL, FactorLp)) else validate_array(L, "L", optional=True)
        self.d = validate_float_or_iterable_numerical(d, "d", optional=True, positive=True)
        self.initial_value = validate_array(initial_value, "initial_value", optional=True)
        self.optimizer = validate_string(optimizer, "optimizer", choices={"adam", "advi", "L-BFGS-B"})
        self.n_iter = validate_positive_int(n_iter, "n_iter")
        self.init_learn_rate = validate_positive_float(init_learn_rate, "init_learn_rate")
        self.predictor_with_uncertainty = validate_bool(predictor_with_uncertainty, "predictor_with_uncertainty")
        self.jit = validate_bool(jit, "jit")
        self.check_rank = validate_bool(check_rank, "check_rank", optional=True)
        self.x = None
        self.pre_transformation = None
        self.pre_transformation_std = None
        self.lbfgsb_options = None      # overrides of inference.LBFGSB_OPTIONS
        self.implicit_factor = True     # False: materialise L = K Lp^-T on the device like the reference does
        self._fit = None                # mln_fit handle holding Lp and L

    def __str__(self):
        return self.__repr__()

    def __repr__(self):
        def s(v):
            if v is None:
                return "None"
            return f"<array {tuple(v.shape)}>" if hasattr(v, "shape") else str(v)
        keys = ("n_landmarks", "rank", "gp_type", "jitter", "d", "mu", "ls", "cov_func", "landmarks", "Lp", "L",
                "nn_distances", "initial_value", "optimizer")
        return self.__class__.__name__ + "(" + ", ".join(f"{k}={s(getattr(self, k, None))}" for k in keys) + ")"

    def __call__(self, x=None):
        return self.fit_predict(x=x)

    # -- x -------------------------------------------------------------------------------------------
    def set_x(self, x):
        """reference base_model.py:176-213 (identity check: a second, different x raises)."""
        if self.x is not None and x is not None and self.x is not x:
            raise ValueError("self.x has been set already, but is not equal to the argument x.")
        if self.x is None and x is None:
            raise ValueError("Required argument x is missing and self.x has not been set.")
        if x is None:
            x = self.x
        x = validate_array(x, "x")
        self.x = x if isinstance(x, _lib.DeviceArray) else ensure_2d(x)   # HBM-resident x is used in place
        return self.x

    # -- lazy attribute pipeline ---------------------------------------------------------------------
    def _prepare_attribute(self, attribute):
        """reference base_model.py:433-446."""
        if getattr(self, attribute) is not None:
            return
        value = getattr(self, "_compute_" + attribute)()
        setattr(self, attribute, value)
        # which values the estimator derived itself (as opposed to ones the user passed in or assigned later: those are
        # different objects): only a derived value may ever be recomputed behind the user's back (run_inference)
        self.__dict__.setdefault("_derived_attributes", {})[attribute] = value

    def _is_derived(self, attribute):
        held = self.__dict__.get("_derived_attributes", {}).get(attribute)
        return held is not None and held is getattr(self, attribute)

    def _compute_n_landmarks(self):
        return compute_n_landmarks(self.gp_type, self.x.shape[0], self.landmarks)

    def _compute_rank(self):
        return compute_rank(self.gp_type)

    def _compute_gp_type(self):
        return compute_gp_type(self.n_landmarks, self.rank, self.x.shape[0])

    def _seed(self):
        return self.random_state if self.random_state is not None else DEFAULT_RANDOM_SEED

    def _require_single_process(self, what):
        from .distributed import current
        if current().world_size > 1:
            raise NotImplementedError(
                f"{what} need all cells; when cells are sharded across ranks pass `{what}=` explicitly "
                "(replicated landmarks / ls_time, this rank's nn_distances).")

    def _all_cells(self):
        """Cell-sharded fit: the cells of ALL ranks in rank order (host array, gathered once over the host
        communicator) and the global index of this rank's first cell.  Single rank: (x, 0).  Only the two inputs that
        need every cell -- nearest-neighbour distances and k-means landmarks (parameters.py:243-291,352-433) -- use it."""
        from .distributed import current
        comm = current()
        x_loc = self.x.to_host() if isinstance(self.x, _lib.DeviceArray) else np.ascontiguousarray(self.x, dtype=np.float64)
        if comm.world_size == 1:
            return x_loc, 0
        cached = getattr(self, "_x_all", None)
        if cached is None:
            parts = comm.host.allgather(x_loc)
            lo = int(sum(p.shape[0] for p in parts[:comm.rank]))
            cached = self._x_all = (np.ascontiguousarray(np.concatenate(parts, axis=0)), lo)
        return cached

    # -- one HBM copy of the cells for the steps before the fit ---------------------------------------------
    DEVICE_X_MIN_BYTES = 1 << 26      # below this the three uploads (1-NN, k-means, fit) cost less than the bookkeeping

    def _x_on_device(self):
        """The cells in HBM: uploaded ONCE for the 1-NN search, the k-means landmarks and the fit instead of once per
        step (single rank, host x of at least DEVICE_X_MIN_BYTES); else self.x itself.  Released by
        _release_x_on_device() when prepare_inference() is through."""
        from .distributed import current
        if isinstance(self.x, _lib.DeviceArray) or self.x is None:
            return self.x
        held = self.__dict__.get("_x_dev")
        if held is not None and held[0] is self.x and held[1].ptr is not None:
            return held[1]
        if current().world_size > 1 or self.x.ndim != 2 or self.x.nbytes < self.DEVICE_X_MIN_BYTES:
            return self.x
        dev = _lib.default_context().to_device(np.ascontiguousarray(self.x, dtype=np.float64))
        self._x_dev = (self.x, dev)
        return dev

    def _x_for_fit(self):
        """The HBM copy if an earlier step made one, else self.x (a host array is uploaded by the fit itself, in chunks
        under its first kernels)."""
        held = self.__dict__.get("_x_dev")
        if held is not None and held[0] is self.x and held[1].ptr is not None:
            return held[1]
        return self.x if isinstance(self.x, _lib.DeviceArray) else np.ascontiguousarray(self.x)

    def _release_x_on_device(self):
        held = self.__dict__.pop("_x_dev", None)
        if held is not None:
            held[1].free()

    # the HBM copy is a handle of this process, not state: copies and pickles of an estimator never carry it
    _TRANSIENT = ("_x_dev", "_x_all")

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in self._TRANSIENT}

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in self._TRANSIENT:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def _compute_landmarks(self, ctx=None):
        from .distributed import current
        comm = current()
        n = self._n_cells_global()
        if n > 100 * self.n_landmarks and n > 1e6:
            logger.info(f"Large number of {n:,} cells and small number of {self.n_landmarks:,} landmarks. Consider "
                        "computing k-means on a subset of cells and passing the results as 'landmarks'.")
        if comm.world_size == 1:
            x = self.x
            if not isinstance(x, _lib.DeviceArray) and self.n_landmarks and self.n_landmarks < x.shape[0] and \
                    landmarks_backend(x.shape[0], x.shape[1], self.n_landmarks) == "hip":
                x = self._x_on_device()
            return compute_landmarks(x, self.gp_type, n_landmarks=self.n_landmarks, random_state=self._seed(), ctx=ctx)
        # replicated input: rank 0 clusters the gathered cells, every rank receives the same bits
        x_all, _ = self._all_cells()
        lm = compute_landmarks(x_all, self.gp_type, n_landmarks=self.n_landmarks, random_state=self._seed()) \
            if comm.rank == 0 else None
        return comm.broadcast(None if lm is None else np.ascontiguousarray(lm, dtype=np.float64), src=0)

    def _compute_nn_distances(self):
        logger.info("Computing nearest neighbor distances.")
        from .distributed import current
        if current().world_size == 1:
            return validate_nn_distances(compute_nn_distances(self._x_on_device(), seed=self._seed()))
        x_all, lo = self._all_cells()
        n_loc = self.x.shape[0]
        # this rank's cells against the cells of all ranks, the pair (i, lo + i) excluded
        nn = _lib.default_context().nn_distances(np.ascontiguousarray(x_all[lo:lo + n_loc]), x_all, self_offset=lo)
        return validate_nn_distances_sharded(nn, current())

    def _compute_ls(self):
        return compute_ls(self.nn_distances) * self.ls_factor

    def _compute_cov_func(self):
        cov_func = compute_cov_func(self.cov_func_curry, self.ls)
        logger.info("Using covariance function %s.", str(cov_func))
        return cov_func

    def _device_fit(self):
        """One mln_fit handle computes Lp AND L (parameters.compute_Lp + compute_L of the reference)."""
        if self._fit is not None:
            return self._fit
        ctx = _lib.default_context()
        given_L = self.L if not isinstance(self.L, (FactorL, FactorLp)) else None
        if isinstance(self.L, (FactorL, FactorLp)):
            self._fit = self.L.fit
        elif given_L is None and self.gp_type == GaussianProcessType.FULL_NYSTROEM:
            self._require_single_process("full_nystroem factor")
            logger.info("Computing rank reduction using all cells (full Nystroem).")
            self._fit = _full_decomposition_low_rank(self.x, self.cov_func, rank=self.rank, jitter=self.jitter).fit
        elif given_L is None and self.gp_type == GaussianProcessType.SPARSE_NYSTROEM:
            logger.info("Computing improved Nystroem rank reduction on the landmarks.")
            xin = self._x_for_fit()
            self._fit = _modified_low_rank(xin, self.cov_func, self.landmarks, rank=self.rank,
                                           jitter=self.jitter).fit
        elif given_L is not None:
            Lp = None if self.Lp is None else np.asarray(self.Lp, dtype=np.float64)
            if Lp is None and self.gp_type == GaussianProcessType.FULL:
                Lp = np.asarray(given_L, dtype=np.float64)
            self._fit = _lib.Fit.from_L(ctx, np.asarray(given_L, dtype=np.float64), Lp=Lp)
        else:
            Lp = None if self.Lp is None else np.asarray(self.Lp, dtype=np.float64)
            full = self.gp_type == GaussianProcessType.FULL or self.landmarks is None
            logger.info("Computing Lp.")
            xin = self._x_for_fit()
            # implicit mode: stream K = cov(x, landmarks) and fold Lp^-T into the m-vectors (no n x m
            # triangular solve); the explicit factor is only needed for the diagonal Laplace.
            self._fit = ctx.fit_prepare(self.cov_func.lower(self.x.shape[1]), xin,
                                        None if full else self.landmarks, self.jitter, Lp=Lp,
                                        implicit=self.implicit_factor and not self.predictor_with_uncertainty)
        return self._fit

    def _compute_Lp(self):
        fit = self._device_fit()
        return FactorLp(fit) if getattr(fit, "_has_lp", True) else None

    def _compute_L(self):
        fit = self._device_fit()
        n_samples = self.x.shape[0]
        n_landmarks = n_samples if self.landmarks is None else self.landmarks.shape[0]
        if self.gp_type in (GaussianProcessType.SPARSE_NYSTROEM, GaussianProcessType.FULL_NYSTROEM) \
                and fit.m > self.rank * RANK_FRACTION_THRESHOLD * n_landmarks:          # base_model.py:333-342
            logger.warning(f"Shallow rank reduction from {n_landmarks:,} to {fit.m:,} indicates underrepresentation "
                           "by landmarks. Consider increasing n_landmarks!")
        # base_model.py:344-355: on request, or automatically for a sparse-Cholesky fit with more than 10 cells per
        # landmark.  Log-only in the reference too (an n x m SVD there; here the m x m Gram over all cells of all ranks
        # and its eigenvalues on the device: ~3 s at 1e6 x 5000, so timed runs pass check_rank=False).
        if self.check_rank or (self.check_rank is None and self.gp_type == GaussianProcessType.SPARSE_CHOLESKY
                               and SAMPLE_LANDMARK_RATIO * n_landmarks < self._n_cells_global()):
            from .util import test_rank
            logger.info(f"Estimating approximation accuracy since {n_samples:,} samples are more than "
                        f"{SAMPLE_LANDMARK_RATIO} x {n_landmarks:,} landmarks.")
            test_rank(FactorL(fit), threshold=RANK_FRACTION_THRESHOLD)
        logger.info(f"Using rank {fit.m:,} covariance representation.")
        return FactorLp(fit) if self.gp_type == GaussianProcessType.FULL and fit.m == fit.n and \
            getattr(fit, "_has_lp", True) else FactorL(fit)

    def _n_cells_global(self):
        """Cells over all ranks (the rank diagnostic's trigger must be the same decision on every rank)."""
        from .distributed import current
        return current().global_count(self.x.shape[0])

    def validate_parameter(self):
        validate_params(self.rank, self.gp_type, self.x.shape[0], self.n_landmarks, self.landmarks)

    # -- optimiser switch (reference base_model.py:371-431) ----------------------------------------------
    def _run_inference(self):
        logger.info("Running inference using %s.", self.optimizer)
        if self.optimizer == "adam":
            results = minimize_adam(self.loss_func, self.initial_value, n_iter=self.n_iter,
                                    init_learn_rate=self.init_learn_rate, jit=self.jit)
            self.pre_transformation = results.pre_transformation
            self.pre_transformation_std = None
            self.opt_state = results.opt_state
            self.losses = results.losses
        elif self.optimizer == "advi":
            run_advi()
        elif self.optimizer == "L-BFGS-B":
            results = minimize_lbfgsb(self.loss_func, self.initial_value, jit=self.jit, options=self.lbfgsb_options)
            self.pre_transformation = results.pre_transformation
            self.pre_transformation_std = None
            self.opt_state = results.opt_state
            self.losses = [results.loss]
        else:
            raise ValueError(f"Unknown optimizer {self.optimizer}.")
        if self.optimizer != "advi" and self.predictor_with_uncertainty and self.pre_transformation_std is None:
            logger.info("Computing Laplace approximation for posterior uncertainty.")
            self.pre_transformation_std = compute_laplace_std(self.loss_func, self.pre_transformation, jit=self.jit)
