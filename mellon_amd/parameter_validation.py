"""Consistency checks between rank / gp_type / landmarks (mellon/parameter_validation.py)."""
import logging

import numpy as np

from .base_cov import Covariance
from .util import GaussianProcessType
from .validation import validate_float_or_int, validate_positive_int

logger = logging.getLogger("mellon")
_NYS = (GaussianProcessType.FULL_NYSTROEM, GaussianProcessType.SPARSE_NYSTROEM)


def validate_landmark_params(n_landmarks, landmarks):
    """reference parameter_validation.py:13-33."""
    if landmarks is not None and n_landmarks != landmarks.shape[0]:
        raise ValueError(
            f"There are {landmarks.shape[0]:,} landmarks specified but n_landmarks={n_landmarks:,}. "
            "Please omit specifying n_landmarks if landmarks are given.")


def validate_rank_params(gp_type, n_samples, rank, n_landmarks):
    """reference parameter_validation.py:36-100."""
    bound = n_landmarks if gp_type in (GaussianProcessType.SPARSE_CHOLESKY, GaussianProcessType.SPARSE_NYSTROEM) \
        else n_samples
    full = ((type(rank) is int and gp_type is not None and gp_type != GaussianProcessType.FIXED and rank >= bound)
            or (type(rank) is float and rank >= 1.0) or rank == 0)
    if full:
        if gp_type == GaussianProcessType.FULL_NYSTROEM:
            raise ValueError(f"Gaussian Process type {gp_type} requires fractional 0 < rank < 1 or integer "
                             f"0 < rank < {n_samples:,} (number of cells) but the actual rank is {rank}.")
        if gp_type == GaussianProcessType.SPARSE_NYSTROEM:
            raise ValueError(f"Gaussian Process type {gp_type} requires fractional 0 < rank < 1 or integer "
                             f"0 < rank < {n_landmarks:,} (number of landmakrs) but the actual rank is {rank}.")
    elif gp_type not in _NYS:
        raise ValueError(f"Given rank {rank} indicates Nyström rank reduction. "
                         f"But the Gaussian Process type is set to {gp_type}.")


def validate_gp_type(gp_type, n_samples, n_landmarks):
    """reference parameter_validation.py:103-150."""
    if gp_type in (GaussianProcessType.FULL, GaussianProcessType.FULL_NYSTROEM) \
            and n_landmarks != 0 and n_landmarks < n_samples:
        raise ValueError(
            f"Gaussian Process type {gp_type} but n_landmarks={n_landmarks:,} is smaller "
            f"than the number of cells {n_samples:,}. Omit n_landmarks or set it to 0 to use "
            "a non-sparse Gaussian Process or omit gp_type to use a sparse one.")
    if gp_type in (GaussianProcessType.SPARSE_CHOLESKY, GaussianProcessType.SPARSE_NYSTROEM):
        if n_landmarks == 0:
            raise ValueError(
                f"Gaussian Process type {gp_type} but n_landmarks=0. Set n_landmarks to a number smaller "
                f"than the number of cells {n_samples:,} to use a sparse Gaussian Process or omit gp_type.")
        if n_landmarks >= n_samples:
            raise ValueError(
                f"Gaussian Process type {gp_type} but n_landmarks={n_landmarks:,} is larger or equal the "
                f"number of cells {n_samples:,}. Reduce the number of landmarks or omit gp_type.")


def validate_params(rank, gp_type, n_samples, n_landmarks, landmarks):
    """reference parameter_validation.py:153-192."""
    n_landmarks = validate_positive_int(n_landmarks, "n_landmarks")
    rank = validate_float_or_int(rank, "rank")
    if not isinstance(gp_type, GaussianProcessType):
        raise ValueError(f"gp_type needs to be a mellon.util.GaussianProcessType but is a {type(gp_type)} instead.")
    validate_landmark_params(n_landmarks, landmarks)
    if n_landmarks > n_samples and gp_type != GaussianProcessType.FIXED:
        logger.warning(f"n_landmarks={n_landmarks:,} is larger than the number of cells {n_samples:,}.")
    validate_gp_type(gp_type, n_samples, n_landmarks)
    validate_rank_params(gp_type, n_samples, rank, n_landmarks)


def validate_cov_func_curry(cov_func_curry, cov_func, param_name):
    """reference parameter_validation.py:195-229."""
    if cov_func_curry is None and cov_func is None:
        raise ValueError("At least one of 'cov_func_curry' and 'cov_func' must not be None")
    if cov_func_curry is not None and not (isinstance(cov_func_curry, type) and issubclass(cov_func_curry, Covariance)):
        if not callable(cov_func_curry):
            raise ValueError(f"'{param_name}' must be a subclass of mellon.Covariance")
    return cov_func_curry


def validate_cov_func(cov_func, param_name, optional=False):
    """reference parameter_validation.py:232-279."""
    if cov_func is None and optional:
        return None
    if not isinstance(cov_func, Covariance):
        raise ValueError(f"'{param_name}' must be an instance of a subclass of mellon.Covariance")
    return cov_func


def validate_normalize_parameter(normalize, unique_times):
    """reference parameter_validation.py:266-280: a dict must name every time point, a list / array must have one
    entry per time point (earliest to latest)."""
    if isinstance(normalize, dict):
        absent = [t for t in unique_times if t.item() not in normalize]
        if absent:
            raise ValueError(f"Missing time point(s) in normalization dictionary: {absent}")
    elif isinstance(normalize, (list, np.ndarray)) and len(normalize) != len(unique_times):
        raise ValueError("Length of the normalize list or array must match the number of unique time points.")
