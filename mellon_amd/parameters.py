"""Parameter heuristics and wiring (mellon/parameters.py).  Scalar heuristics are host NumPy
(O(n)); Lp / L / initial_value run on the device through libmellon_hip.so."""
import logging

import numpy as np

from . import _lib
from .decomposition import (DEFAULT_RANK, FactorL, FactorLp, _full_decomposition_low_rank, _full_rank,
                            _modified_low_rank, _standard_low_rank)
from .util import DEFAULT_JITTER, GaussianProcessType, ensure_2d, mle
from .validation import (validate_array, validate_float_or_int, validate_k, validate_positive_float,
                         validate_positive_int, validate_time_x)

DEFAULT_N_LANDMARKS = 5000     # reference parameters.py:53
DEFAULT_RANDOM_SEED = 42       # reference parameters.py:54
DEFAULT_SIGMA = 0

logger = logging.getLogger("mellon")


def compute_rank(gp_type):
    """reference parameters.py:88-115."""
    if gp_type in (GaussianProcessType.FULL_NYSTROEM, GaussianProcessType.SPARSE_NYSTROEM):
        return DEFAULT_RANK
    return 1.0


def compute_n_landmarks(gp_type, n_samples, landmarks):
    """reference parameters.py:118-172."""
    if landmarks is not None:
        return landmarks.shape[0]
    if gp_type is None or gp_type == GaussianProcessType.FIXED:
        return min(n_samples, DEFAULT_N_LANDMARKS)
    if gp_type in (GaussianProcessType.FULL, GaussianProcessType.FULL_NYSTROEM):
        return n_samples
    if gp_type in (GaussianProcessType.SPARSE_CHOLESKY, GaussianProcessType.SPARSE_NYSTROEM):
        if n_samples <= DEFAULT_N_LANDMARKS:
            logger.warning(
                f"Gaussian Process type {gp_type} and default number of landmarks {DEFAULT_N_LANDMARKS:,} "
                f"< number of cells {n_samples:,}. Reduce n_landmarks below the number of cells to use {gp_type}.")
        return DEFAULT_N_LANDMARKS
    n_landmarks = min(n_samples, DEFAULT_N_LANDMARKS)
    logger.warning(f"Unknown Gaussian Process type {gp_type}, using default n_landmarks={n_landmarks:,}.")
    return n_landmarks


def _indicates_full_rank(rank, bound):
    return (rank is None or (isinstance(rank, int) and rank >= bound)
            or (isinstance(rank, float) and rank >= 1.0) or rank == 0)


def compute_gp_type(n_landmarks, rank, n_samples):
    """reference parameters.py:175-240 (decision table pinned by its tests/test_parameters.py:271-290)."""
    rank = validate_float_or_int(rank, "rank", optional=True)
    n_landmarks = validate_positive_int(n_landmarks, "n_landmarks")
    n_samples = validate_positive_int(n_samples, "n_samples")
    if n_landmarks == 0 or n_landmarks >= n_samples:
        if _indicates_full_rank(rank, n_samples):
            logger.info(f"Using non-sparse Gaussian Process since n_landmarks ({n_landmarks:,}) >= "
                        f"n_samples ({n_samples:,}) and rank = {rank}.")
            return GaussianProcessType.FULL
        return GaussianProcessType.FULL_NYSTROEM
    if _indicates_full_rank(rank, n_landmarks):
        logger.info(f"Using sparse Gaussian Process since n_landmarks ({n_landmarks:,}) < "
                    f"n_samples ({n_samples:,}) and rank = {rank}.")
        return GaussianProcessType.SPARSE_CHOLESKY
    return GaussianProcessType.SPARSE_NYSTROEM


KMEANS_DEVICE_THRESHOLD = 2e8   # n * n_landmarks above which k-means runs on the device
# Seeding of the device k-means: "device" (the library's own draws: 0.08 s at 1e6 x 50 -> 5000) or "sklearn" (the cells
# sklearn's own k-means++ picks for the same random_state -- the landmarks of the default call are then comparable with the
# reference's at any size; 2 s at that size).  The environment variable MELLON_AMD_KMEANS_INIT overrides.
KMEANS_DEVICE_INIT = "device"


def landmarks_backend(n, d, n_landmarks):
    """Where compute_landmarks runs by default: "hip" when n * n_landmarks exceeds KMEANS_DEVICE_THRESHOLD and d <= 64."""
    return "hip" if (n * n_landmarks > KMEANS_DEVICE_THRESHOLD and d <= 64) else "sklearn"


def compute_landmarks(x, gp_type=None, n_landmarks=DEFAULT_N_LANDMARKS, random_state=DEFAULT_RANDOM_SEED,
                      backend=None, ctx=None):
    """k-means centroids (reference parameters.py:243-291).  backend "sklearn" is the reference's
    own call (bit-identical landmarks); "hip" is k-means++ / Lloyd on the device (mln_kmeans: same
    algorithm family; its seeding is the library's own or, with KMEANS_DEVICE_INIT = "sklearn", sklearn's); None picks "hip" when n * n_landmarks exceeds
    KMEANS_DEVICE_THRESHOLD (where sklearn takes minutes) and d <= 64.  x may be HBM-resident (a DeviceArray);
    ctx: the device context to run on (default: the calling thread's)."""
    if n_landmarks == 0:
        return None
    on_device = isinstance(x, _lib.DeviceArray)
    x = x if on_device else ensure_2d(x)
    n = x.shape[0]
    assert n_landmarks > 1, "n_landmarks musst be larger 1 or euqual to 0"
    if n_landmarks >= n:
        if gp_type == GaussianProcessType.FIXED:
            logger.info(f"Using all {n:,} datapoints as landmarks.")
            return x
        return None
    if backend is None:
        backend = landmarks_backend(n, x.shape[1], n_landmarks)      # (HBM-resident cells follow the same rule: sklearn gets a host copy)
    logger.info(f"Computing {n_landmarks:,} landmarks with k-means clustering "
                f"(random_state={random_state}, backend={backend}).")
    if backend == "hip":
        import os
        init = os.environ.get("MELLON_AMD_KMEANS_INIT", KMEANS_DEVICE_INIT)
        return (ctx or _lib.default_context()).kmeans(x if on_device else np.ascontiguousarray(x, dtype=np.float64), n_landmarks,
                                                      seed=random_state if random_state is not None else DEFAULT_RANDOM_SEED,
                                                      init=init)
    if on_device:
        x = x.to_host()
    from sklearn.cluster import k_means
    return k_means(x, n_landmarks, n_init=1, random_state=random_state)[0]


def compute_landmarks_rescale_time(x, ls, ls_time, times=None, n_landmarks=DEFAULT_N_LANDMARKS,
                                   random_state=DEFAULT_RANDOM_SEED):
    """reference parameters.py:294-349: k-means with the time column rescaled by ls / ls_time."""
    if n_landmarks == 0:
        return None
    ls = validate_positive_float(ls, "ls")
    ls_time = validate_positive_float(ls_time, "ls_time")
    x = np.array(validate_time_x(x, times), dtype=np.float64)
    factor = ls / ls_time
    x[:, -1] *= factor
    landmarks = compute_landmarks(x, n_landmarks=n_landmarks, random_state=random_state)
    if landmarks is not None:
        landmarks = np.array(landmarks)
        landmarks[:, -1] /= factor
    return landmarks


def compute_distances(x, k, seed=DEFAULT_RANDOM_SEED):
    """Distances to the k nearest neighbours.  The reference uses pynndescent (approximate,
    parameters.py:352-405); this is the exact Euclidean answer from a space-partitioning tree."""
    x = ensure_2d(validate_array(x, "x"))
    n = x.shape[0]
    if n == 0:
        raise ValueError("Input data x is empty.")
    validate_k(k, n)
    from sklearn.neighbors import BallTree, KDTree
    tree = (KDTree if x.shape[1] <= 20 else BallTree)(x)
    dist, _ = tree.query(x, k=k + 1)
    return dist[:, 1:]


def compute_nn_distances(x, seed=DEFAULT_RANDOM_SEED):
    """reference parameters.py:408-433 -- exact, brute force on the device (mln_nn_distances);
    the reference's pynndescent search is approximate and `seed` only matters there."""
    if isinstance(x, _lib.DeviceArray):          # HBM-resident cells (validated when they were uploaded)
        validate_k(1, x.shape[0])
        return _lib.default_context().nn_distances(x)
    x = ensure_2d(validate_array(x, "x"))
    if x.shape[0] == 0:
        raise ValueError("Input data x is empty.")
    validate_k(1, x.shape[0])
    return _lib.default_context().nn_distances(np.ascontiguousarray(x, dtype=np.float64))


def _target_cell_count(normalize, t, average, unique_times):
    """reference parameters.py:436-441: True -> the average count per time point, dict -> by time value,
    list / array -> by position among the sorted unique time points."""
    if isinstance(normalize, bool):
        return average
    if isinstance(normalize, dict):
        return normalize[t.item()]
    return normalize[unique_times.tolist().index(t)]


def compute_nn_distances_within_time_points(x, times=None, d=None, normalize=False, local=None):
    """reference parameters.py:444-531.  `local=(lo, n_local)`: x holds the cells of ALL ranks and only the rows
    [lo, lo + n_local) -- this rank's shard -- are searched for (against every cell of their time point) and returned;
    counts per time point and their average are then the global ones by construction."""
    from .parameter_validation import validate_normalize_parameter
    from .validation import validate_float_or_iterable_numerical
    x = validate_time_x(x, times)
    unique_times = np.unique(x[:, -1])
    n_cells = x.shape[0]
    lo, n_loc = (0, n_cells) if local is None else (int(local[0]), int(local[1]))
    nn = np.empty(n_loc)
    av = n_cells / len(unique_times)
    validate_normalize_parameter(normalize, unique_times)
    normalizing = normalize is not False and normalize is not None
    if normalizing:
        d = validate_float_or_iterable_numerical(d, "d", optional=False, positive=True)
        if np.ndim(d) > 0 and len(d) != n_loc:
            raise ValueError(f"If `d` (length={len(d):,}) is a vector then it needs to have one value "
                             f"per cell in x (x.shape[0]={n_loc:,}).")
        logger.info("Normalizing nearest neighbor distances correcting sampling bias for "
                    f"{len(unique_times):,} different time points.")
    own = np.zeros(n_cells, dtype=bool)
    own[lo:lo + n_loc] = True
    for t in unique_times:
        mask = x[:, -1] == t
        n_t = int(mask.sum())
        if n_t < 2:
            raise ValueError(
                f"Insufficient data: Only {n_t} sample(s) found at time point {t}. "
                "Nearest neighbors cannot be computed with less than two samples per time point.")
        mine = mask & own
        if not mine.any():
            continue
        if local is None:
            nn_t = compute_nn_distances(x[mask, :-1])
        else:
            # this rank's cells of the time point are a contiguous run of the time point's cells (order preserved)
            before = int((mask[:lo]).sum())
            x_t = np.ascontiguousarray(x[mask, :-1], dtype=np.float64)
            k_t = int(mine.sum())
            nn_t = _lib.default_context().nn_distances(np.ascontiguousarray(x_t[before:before + k_t]), x_t,
                                                       self_offset=before)
        sel = mine[lo:lo + n_loc]
        if normalizing:
            target = _target_cell_count(normalize, t, av, unique_times)
            dd = np.asarray(d, dtype=np.float64)
            nn_t = (n_t / target) ** (1 / dd if dd.ndim == 0 else 1 / dd[sel]) * nn_t
        nn[sel] = nn_t
    return nn


def compute_d(x):
    """reference parameters.py:534-549: the embedding dimensionality."""
    x = np.asarray(x) if not isinstance(x, _lib.DeviceArray) else x
    return 1 if len(x.shape) < 2 else x.shape[1]


def compute_mu(nn_distances, d):
    """reference parameters.py:586-599 (1st percentile, linear interpolation, minus 10) -- over
    the GLOBAL cell set when cells are sharded across ranks."""
    from .distributed import current
    return current().global_quantile(mle(np.asarray(nn_distances, dtype=np.float64), d), 0.01) - 10


def compute_ls(nn_distances):
    """reference parameters.py:602-613 (global mean of log nn when sharded)."""
    from .distributed import current
    from .util import log_nn
    return float(np.exp(current().global_mean(log_nn(nn_distances)) + 3.0))


def compute_cov_func(cov_func_curry, ls, ls_time=None):
    """reference parameters.py:616-645."""
    if ls_time is not None:
        return cov_func_curry(ls=ls, active_dims=slice(None, -1)) * cov_func_curry(ls=ls_time, active_dims=-1)
    return cov_func_curry(ls=ls)


def compute_average_cell_count(x, normalize):
    """reference parameters.py:927-969 (cells and time points of ALL ranks when sharded)."""
    from .distributed import current
    comm = current()
    local_times = np.unique(x[:, -1])
    if comm.world_size > 1:
        n_times = np.unique(np.concatenate(comm.host.allgather(local_times))).shape[0]
        n_cells = comm.global_count(x.shape[0])
    else:
        n_times, n_cells = local_times.shape[0], x.shape[0]
    if normalize is None or isinstance(normalize, bool):
        return n_cells / n_times
    if isinstance(normalize, dict):
        return sum(normalize.values()) / n_times
    if isinstance(normalize, (list, np.ndarray)):
        return float(np.sum(np.asarray(normalize)) / len(normalize))
    raise ValueError(f"Unrecognized type for 'normalize': {type(normalize)}")


def compute_Lp(x, cov_func, gp_type=None, landmarks=None, sigma=DEFAULT_SIGMA, jitter=DEFAULT_JITTER):
    """reference parameters.py:648-714.  Returns a device-resident FactorLp (or None for Nystroem)."""
    n_samples = x.shape[0]
    n_landmarks = n_samples if landmarks is None else ensure_2d(landmarks).shape[0]
    gp_type = GaussianProcessType.from_string(gp_type, optional=True)
    if gp_type is None:
        gp_type = compute_gp_type(n_landmarks, 1.0, n_samples)
    if gp_type in (GaussianProcessType.FULL_NYSTROEM, GaussianProcessType.SPARSE_NYSTROEM):
        return None
    if gp_type == GaussianProcessType.FULL:
        logger.info("Computing Lp.")
        return _full_rank(x, cov_func, sigma=sigma, jitter=jitter)
    if gp_type in (GaussianProcessType.SPARSE_CHOLESKY, GaussianProcessType.FIXED):
        return _full_rank(landmarks, cov_func, sigma=sigma, jitter=jitter)
    raise ValueError(f"Unknown Gaussian Process type {gp_type}.")


def compute_L(x, cov_func, gp_type=None, landmarks=None, Lp=None, rank=None, sigma=DEFAULT_SIGMA,
              jitter=DEFAULT_JITTER):
    """reference parameters.py:783-874.  Returns a device-resident factor."""
    n_samples = x.shape[0]
    n_landmarks = n_samples if landmarks is None else ensure_2d(landmarks).shape[0]
    gp_type = GaussianProcessType.from_string(gp_type, optional=True)
    if gp_type is None:
        gp_type = compute_gp_type(n_landmarks, rank, n_samples)
    if Lp is not None and tuple(Lp.shape) != (n_landmarks, n_landmarks):
        raise ValueError(f"Wrong shape of Lp {tuple(Lp.shape)}; expected {(n_landmarks, n_landmarks)}.")
    if gp_type == GaussianProcessType.FULL:
        if Lp is None:
            return _full_rank(x, cov_func, sigma=sigma, jitter=jitter)
        return Lp                                           # parameters.py:847-850
    if gp_type == GaussianProcessType.FULL_NYSTROEM:
        return _full_decomposition_low_rank(x, cov_func, rank=rank, sigma=sigma, jitter=jitter)
    if gp_type in (GaussianProcessType.SPARSE_CHOLESKY, GaussianProcessType.FIXED):
        return _standard_low_rank(x, cov_func, landmarks, Lp=Lp, sigma=sigma, jitter=jitter)
    if gp_type == GaussianProcessType.SPARSE_NYSTROEM:
        return _modified_low_rank(x, cov_func, landmarks, rank=rank, sigma=sigma, jitter=jitter)
    raise ValueError(f"Unknown Gaussian Process type {gp_type}.")


def _fit_of(L):
    if isinstance(L, (FactorL, FactorLp)):
        return L.fit
    return _lib.Fit.from_L(_lib.default_context(), np.asarray(L, dtype=np.float64))


RIDGE_ROWS_PER_LANDMARK = 6     # cells of the first preconditioner's Gram per landmark (round 3, tools/solver_sweep.py, five seeds at C3:
                                # 6 m: 192 ms, 8 m: 193, 12 m: 198, 24 m: 207 -- the subsample phase and the rebuild made the first factor's
                                # accuracy matter less than its Gram time)


def ridge_row_stride(n_local, m, with_offset=False):
    """Cells used for the Ridge / preconditioner Gram: every k-th cell (by GLOBAL index) such that ~12 m cells remain
    (all cells when n <= 24 m).  The Gram only seeds and preconditions a strictly convex solve, so
    the subsample changes the iteration count (measured: not at all down to 2 m rows), never the
    optimum; `row_stride=1` reproduces the reference's exact Ridge on all cells.
    with_offset: also return the global index of this rank's first cell."""
    import os
    from .distributed import current
    per_m = RIDGE_ROWS_PER_LANDMARK
    if os.environ.get("MELLON_AMD_EXPERIMENTAL", "0") not in ("", "0"):       # (sweep knob: csrc/mln_options.h)
        per_m = int(os.environ.get("MELLON_AMD_RIDGE_ROWS_PER_M", per_m))
    offset, n_global = current().global_offset(n_local)
    stride = max(1, n_global // (per_m * int(m))) if n_global > 2 * per_m * int(m) else 1
    return (stride, offset) if with_offset else stride


def compute_initial_value(nn_distances, d, mu, L, row_stride=None, target=None):
    """Ridge(alpha=1, fit_intercept=False) of mle - mu on L (reference parameters.py:877-896),
    solved on the device: (L^T L + I)^-1 L^T t, the Gram taken over every `row_stride`-th cell
    (None = automatic, see ridge_row_stride; 1 = all cells, the reference's exact Ridge)."""
    if target is None:
        target = mle(np.asarray(nn_distances, dtype=np.float64), d) - mu
    fit = _fit_of(L)
    auto, offset = ridge_row_stride(fit.n, fit.m, with_offset=True)
    fit.precond_build(auto if row_stride is None else row_stride, offset, force=row_stride is not None)
    return fit.ridge_init(target)


# -- thin helpers over a predictor (parameters.py:59-86) ---------------------------------------------------------
def compute_initial_zeros(x, L):
    return np.zeros((np.shape(x)[0], np.shape(L)[1]))


def compute_initial_ones(x, L):
    return np.ones(np.shape(x)[0])


def compute_time_derivatives(predictor, x, times=None):
    if hasattr(predictor, "time_derivative"):
        return predictor.time_derivative(x, times)
    return np.zeros(np.shape(x)[0])


def compute_density_gradient(predictor, x, times=None):
    if hasattr(predictor, "time_derivative"):
        return predictor.gradient(x, times)
    return predictor.gradient(x)


def compute_density_diffusion(predictor, x, times=None):
    """parameters.py:81-85 evaluates the Hessian log-determinant and returns nothing; the pair is returned here."""
    if hasattr(predictor, "time_derivative"):
        return predictor.hessian_log_determinant(x, times)
    return predictor.hessian_log_determinant(x)
