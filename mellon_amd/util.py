"""Host-side helpers mirroring mellon/util.py (NumPy only; no device code here)."""
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


def mle(nn_distances, d):
    """reference util.py:334-348."""
    return gammaln(d / 2 + 1) - (d / 2) * np.log(np.pi) - d * np.log(nn_distances)


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
