"""Host-side helpers mirroring mellon/util.py (NumPy only; no device code here)."""
import functools
import inspect
import logging
from enum import Enum

import numpy as np
from scipy.special import gammaln

logger = logging.getLogger("mellon")

DEFAULT_JITTER = 1e-6      # reference util.py:48
DEFAULT_RANK_TOL = 5e-1    # reference util.py:49


def ensure_2d(X):
    """reference util.py:135-147: a 1-D array becomes n x 1."""
    X = np.asarray(X)
    return np.atleast_2d(X.T).T


def select_active_dims(x, active_dims):
    """reference util.py:150-171."""
    if active_dims is not None:
        if np.isscalar(active_dims):
            active_dims = [active_dims]
        x = x[..., active_dims]
    return x


def compose_active_dims(cols, active_dims):
    """Column indices that survive one more `active_dims` selection (same rules as
    select_active_dims, applied to an index vector instead of the data)."""
    return np.atleast_1d(select_active_dims(np.asarray(cols), active_dims))


def log_nn(nn_distances):
    """log of the nearest-neighbour distances, once per array: ls (parameters.py:613), mle / mu (util.py:348, parameters.py:599)
    and the likelihood constants (inference.py:83-85) all start from it -- three passes of np.log over n cells otherwise, the
    first of them on the critical path of a fit (nothing can be sent to the device before ls is known).  Cached on the array's
    identity, size and three of its values -- WITHIN one fit: every prepare_inference() starts a new scope (log_nn_new_fit), so
    that a second fit of the same array pays for its logarithms like the first (a benchmark loop over one array would
    otherwise time steps that skip them).  (Splitting the pass over threads was measured: 18 ms against np.log's 2.2 ms
    at 1e6 cells in an 8-core container -- first-touch page faults of the output contend.)"""
    r = np.asarray(nn_distances, dtype=np.float64)
    if r.ndim != 1 or r.size < 100_000:
        return np.log(r)
    key = (log_nn._scope, id(nn_distances), r.ctypes.data, r.size, float(r[0]), float(r[-1]), float(r[r.size // 2]))
    hit = log_nn._cache
    if hit is not None and hit[0] == key:
        return hit[1]
    out = np.log(r)
    out.setflags(write=False)
    log_nn._cache = (key, out)
    return out


log_nn._cache = None
log_nn._scope = 0


def log_nn_new_fit():
    """A new fit begins: the logarithms cached for the previous one are not reused."""
    log_nn._scope += 1
    log_nn._cache = None


def mle(nn_distances, d):
    """reference util.py:334-348."""
    return gammaln(d / 2 + 1) - (d / 2) * np.log(np.pi) - d * log_nn(nn_distances)


def _None_to_str(v):
    return "None" if v is None else v


def _str_to_None(v):
    return None if isinstance(v, str) and v == "None" else v


def make_serializable(x):
    """Wire format of reference util.py:69-92 (arrays are tagged "jax.numpy" there)."""
    if isinstance(x, np.ndarray):
        return {"type": "jax.numpy", "data": x.tolist()}
    if isinstance(x, np.integer):
        return int(x)
    if isinstance(x, np.floating):
        return float(x)
    if isinstance(x, slice):
        return {"type": "slice", "data": [_None_to_str(v) for v in (x.start, x.stop, x.step)]}
    if isinstance(x, dict):
        return {"type": "dict", "data": {k: make_serializable(v) for k, v in x.items()}}
    if isinstance(x, set):
        return {"type": "set", "data": [make_serializable(v) for v in x]}
    return _None_to_str(x)


def deserialize(s):
    """Inverse of make_serializable (reference util.py:101-132)."""
    if isinstance(s, dict):
        t = s["type"]
        if t == "jax.numpy":
            return np.array(s["data"])
        if t == "slice":
            return slice(*[_str_to_None(v) for v in s["data"]])
        if t == "dict":
            return {k: deserialize(v) for k, v in s["data"].items()}
        if t == "set":
            return {deserialize(v) for v in s["data"]}
        return s
    return _str_to_None(s)


class GaussianProcessType(str, Enum):
    """reference util.py:589-667."""

    FULL = "full"
    FULL_NYSTROEM = "full_nystroem"
    SPARSE_CHOLESKY = "sparse_cholesky"
    SPARSE_NYSTROEM = "sparse_nystroem"
    FIXED = "fixed"

    @staticmethod
    def from_string(s, optional=False):
        if s is None:
            if optional:
                return None
            raise ValueError("Gaussian Process type must be specified but is None.")
        if isinstance(s, GaussianProcessType):
            return s
        try:
            return GaussianProcessType(str(s).lower())
        except ValueError:
            options = [g.value for g in GaussianProcessType]
            close = [o for o in options if str(s).lower() in o or o in str(s).lower()]
            if len(close) == 1:
                logger.warning(f"Gaussian Process type {s} not found. Using closest match {close[0]}.")
                return GaussianProcessType(close[0])
            raise ValueError(f"Gaussian Process type {s} not found. Valid options are {options}.")

    def __str__(self):
        return self.value


def set_verbosity(verbose):
    """reference util.py:539-569."""
    logger.setLevel(logging.INFO if verbose else logging.WARNING)


DEFAULT_RANK_TOL = 5e-1        # util.py:49


def stabilize(A, jitter=DEFAULT_JITTER):
    """A + jitter I (util.py:269-275)."""
    A = np.asarray(A, dtype=np.float64)
    return A + np.eye(A.shape[0]) * jitter


def add_diagonal(A, value):
    """A + value I (util.py:278-293)."""
    A = np.asarray(A, dtype=np.float64)
    return A + np.eye(A.shape[0]) * value


def add_variance(K, M=None, jitter=DEFAULT_JITTER):
    """K + M M^T with the diagonal of M M^T floored at jitter (util.py:296-331)."""
    if M is None:
        return stabilize(K, jitter)
    if np.isscalar(M):
        return add_diagonal(K, max(jitter, M ** 2))
    M = np.asarray(M, dtype=np.float64)
    noise = M @ M.T
    dn = np.diagonal(noise)
    return np.asarray(K, dtype=np.float64) + noise + np.diag(np.where(dn < jitter, jitter - dn, 0.0))


def distance(x, y, eps=1e-12):
    """Pairwise Euclidean distances sqrt(max(|x|^2 - 2 x.y + |y|^2 + eps, 0)) (util.py:351-366), evaluated by the
    device's kernel-matrix pass (the `+ 1e-12` is part of the device arithmetic; other eps are not supported)."""
    if eps != 1e-12:
        raise ValueError("the device distance uses the reference's eps = 1e-12")
    from . import _lib
    x = np.ascontiguousarray(ensure_2d(x), dtype=np.float64)
    y = np.ascontiguousarray(ensure_2d(y), dtype=np.float64)
    return _lib.default_context().pairwise_distance(x, y)


def test_rank(input, tol=DEFAULT_RANK_TOL, threshold=None):
    """Approximate rank of the transformation L: the number of singular values above tol * the largest
    (numpy.linalg.matrix_rank(L, rtol=tol), util.py:429-483).  The singular values are the square roots of the
    eigenvalues of the m x m Gram L^T L, formed and diagonalised on the device."""
    from . import _lib
    from .decomposition import FactorL, FactorLp
    L = input if (hasattr(input, "shape") or isinstance(input, (FactorL, FactorLp))) else getattr(input, "L", input)
    if L is None:
        raise AttributeError("Matrix L is not found in the estimator object. Consider running `.prepare_inference()`.")
    if not hasattr(L, "shape"):
        raise TypeError("Input must be either a matrix or a mellon enstimator with a transformation L.")
    if len(L.shape) != 2:
        raise ValueError("Matrix L must be 2D.")
    # a COUNT of singular values above a threshold needs neither eigenvectors nor eigenvalues: Sturm sequences on the
    # tridiagonalised Gram (csrc/tridiag.hip) -- 0.4 s at 5000 landmarks, where the eigensolver takes seconds
    n_rows = int(L.shape[0])
    if isinstance(L, (FactorL, FactorLp)) and L.fit.handle is not None and not isinstance(L, FactorLp):
        # a cell-sharded factor: the Gram is all-reduced, so this is a COLLECTIVE call (every rank must make it, as the
        # estimator's own check does) and the rank is that of the whole n x m factor
        from .distributed import current
        approx_rank, _ = L.fit.gram_rank(tol)
        n_rows = current().global_count(n_rows)
    else:
        # a plain array is this caller's own matrix: a local diagnostic, as in the reference.  It runs on a context
        # WITHOUT the rank's communicator, so that one rank calling it alone never waits inside a collective.
        Lh = np.asarray(L, dtype=np.float64)
        if Lh.shape[0] < Lh.shape[1]:
            Lh = Lh.T                       # same singular values, smaller Gram
        base = _lib.default_context()
        ctx = base if base.n_ranks == 1 else _lib.Context(base.device)
        fit = _lib.Fit.from_L(ctx, np.ascontiguousarray(Lh))
        try:
            approx_rank, _ = fit.gram_rank(tol)
        finally:
            fit.close()
            if ctx is not base:
                ctx.close()
    max_rank = int(min(n_rows, L.shape[1]))
    rank_fraction = approx_rank / max_rank
    if threshold is not None:
        if rank_fraction > threshold:
            logger.warning(f"High approx. rank fraction ({rank_fraction:.1%}). Consider increasing 'n_landmarks'.")
        else:
            logger.info(f"Rank fraction ({rank_fraction:.1%}, lower is better) is within acceptable range. "
                        "Current settings should provide satisfactory model performance.")
    else:
        print(f"The approx. rank fraction is {rank_fraction:.1%} ({approx_rank:,} of {max_rank:,}). Lower is better.")
    return approx_rank


test_rank.__test__ = False       # a utility of the reference's API, not a pytest case


def make_multi_time_argument(func):
    """Decorator adding an optional `multi_time` argument to a method that takes `time` (util.py:206-265): the method
    runs once per value and the results are stacked along axis 1 (element-wise for tuple results), the layout of the
    reference's `vmap(..., out_axes=1)`."""
    sig = inspect.signature(func)
    new_sig = sig.replace(parameters=list(sig.parameters.values()) + [
        inspect.Parameter("multi_time", inspect.Parameter.POSITIONAL_OR_KEYWORD, default=None)])

    @functools.wraps(func)
    def wrapper(self, *args, **kwargs):
        multi_time = kwargs.pop("multi_time", None)
        if multi_time is None:
            return func(self, *args, **kwargs)
        if kwargs.get("time", None) is not None:
            raise ValueError("Cannot specify both 'time' and 'multi_time' arguments")
        times = np.asarray(multi_time, dtype=np.float64).reshape(-1)
        outs = [func(self, *args, **kwargs, time=float(t)) for t in times]
        if isinstance(outs[0], tuple):
            return tuple(np.stack([o[k] for o in outs], axis=1) for k in range(len(outs[0])))
        return np.stack(outs, axis=1)

    wrapper.__signature__ = new_sig
    return wrapper
