"""Input validators with the error behaviour of mellon/validation.py (TypeError for None /
non-array input, ValueError for shapes and ranges); arrays come back as float64 NumPy."""
import logging

import numpy as np

logger = logging.getLogger("mellon")


def validate_array(iterable, name, optional=False, ndim=None):
    """reference validation.py:302-361."""
    if iterable is None:
        if optional:
            return None
        raise TypeError(f"'{name}' can't be None.")
    from ._lib import DeviceArray
    if isinstance(iterable, DeviceArray):
        return iterable
    if isinstance(iterable, np.ndarray) and iterable.dtype == np.float64:
        array = iterable      # identity is preserved on purpose (set_x compares with `is`)
    else:
        try:
            array = np.asarray(iterable, dtype=np.float64)
        except (TypeError, ValueError):
            raise TypeError(f"'{name}' should be iterable and numeric, got {type(iterable)} instead.")
    if array.ndim == 0 and not np.ndim(iterable) == 0:
        raise TypeError(f"'{name}' should be iterable, got {type(iterable)} instead.")
    if ndim is not None:
        dims = (ndim,) if isinstance(ndim, int) else tuple(ndim)
        if array.ndim not in dims:
            raise ValueError(f"'{name}' must be a {dims}-dimensional array, got {array.ndim}D array instead.")
    return array


def _is_number(v):
    return isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, bool)


def validate_float_or_int(value, param_name, optional=False):
    """reference validation.py:104-150."""
    if value is None and optional:
        return None
    if isinstance(value, (int, np.integer)) and not isinstance(value, bool):
        return int(value)
    try:
        value = float(value)
    except (TypeError, ValueError):
        raise ValueError(f"'{param_name}' should be a float or an int but is {type(value)}")
    if np.isnan(value):
        raise ValueError(f"'{param_name}' should be a non-NaN float number")
    return value


def validate_positive_float(value, param_name, optional=False):
    """reference validation.py:153-201."""
    if value is None and optional:
        return None
    try:
        value = float(value)
    except (TypeError, ValueError):
        raise ValueError(f"'{param_name}' should be a float number but is {type(value)}")
    if value < 0:
        raise ValueError(f"'{param_name}' should be a positive float number")
    if np.isnan(value):
        raise ValueError(f"'{param_name}' should be a non-NaN float number")
    return value


def validate_float(value, param_name, optional=False):
    """reference validation.py:204-259."""
    if value is None:
        if optional:
            return None
        raise ValueError(f"'{param_name}' should be a float number but is None")
    try:
        value = float(value)
    except (TypeError, ValueError):
        raise ValueError(f"'{param_name}' should be a float number but is {type(value)}")
    if np.isnan(value):
        raise ValueError(f"'{param_name}' should be a non-NaN float number")
    return value


def validate_positive_int(value, param_name, optional=False):
    """reference validation.py:262-299 (0 is accepted: n_landmarks=0 means 'full')."""
    if value is None and optional:
        return None
    ok = isinstance(value, (int, np.integer)) and not isinstance(value, bool)
    if not ok and isinstance(value, (float, np.floating)) and float(value).is_integer():
        ok, value = True, int(value)
    if not ok or value < 0:
        raise ValueError(f"'{param_name}' should be a positive integer number")
    return int(value)


def validate_bool(value, name, optional=False):
    """reference validation.py:364-400."""
    if value is None:
        if optional:
            return None
        raise TypeError(f"'{name}' can't be None.")
    if not isinstance(value, (bool, np.bool_)):
        raise TypeError(f"{name} should be of type bool, got {type(value)} instead.")
    return bool(value)


def validate_string(value, name, choices=None):
    """reference validation.py:403-435."""
    if not isinstance(value, str):
        raise TypeError(f"{name} should be of type str, got {type(value)} instead.")
    if choices is not None and value not in choices:
        raise ValueError(f"{name} should be one of {choices}, got '{value}' instead.")
    return value


def validate_float_or_iterable_numerical(value, name, optional=False, positive=False):
    """reference validation.py:438-494."""
    if value is None and optional:
        return None
    if _is_number(value):
        if positive and value < 0:
            raise ValueError(f"{name} should be a non-negative number or array")
        return float(value)
    if hasattr(value, "__iter__") or isinstance(value, np.ndarray):
        arr = np.asarray(value, dtype=np.float64)
        if positive and np.any(arr < 0):
            raise ValueError(f"All elements in {name} should be non-negative")
        return arr if arr.ndim > 0 else float(arr)
    raise TypeError(f"'{name}' should be a number or an iterable of numbers, got {type(value)} instead.")


def validate_1d(x):
    """reference validation.py:497-525."""
    x = np.asarray(x)
    if x.ndim == 0:
        x = x[None]
    if x.ndim != 1:
        raise ValueError("`x` must be exactly 1-dimensional.")
    return x


def validate_nn_distances(nn_distances, optional=False):
    """reference validation.py:528-592: NaN / inf / <= 0 are replaced by the smallest positive
    distance; if every entry is invalid a ValueError is raised."""
    if nn_distances is None:
        if optional:
            return None
        raise ValueError("nn_distances are required but None is given.")
    nn = np.asarray(nn_distances, dtype=np.float64)
    # the common case in two passes: smallest > 0 and largest finite (a NaN anywhere makes either comparison false)
    if nn.ndim == 1 and nn.size and nn.min() > 0 and nn.max() < np.inf:
        return nn
    bad = np.isnan(nn) | np.isinf(nn) | (nn <= 0)
    n_bad = int(bad.sum())
    if n_bad == nn.size:
        raise ValueError(
            f"All {n_bad:,} computed nearest neighbor distances (`nn_distances` attribute) contain invalid "
            "values. Please check the input data.")
    if n_bad:
        logger.warning(
            f"The computed nearest neighbor distances (`nn_distances` attribute) contain {n_bad:,} invalid "
            "values. Setting invalid distances to the minimum positive value found.")
        nn = np.where(~bad, nn, nn[~bad].min())
    return nn


def validate_nn_distances_sharded(nn_distances, comm):
    """validate_nn_distances for a cell-sharded fit: the replacement value is the smallest positive distance over
    the cells of ALL ranks (the reference sees all cells, validation.py:528-592), and "every entry invalid" is
    decided -- and raised -- on every rank together, so no rank walks into a collective alone."""
    if comm is None or comm.world_size == 1:
        return validate_nn_distances(nn_distances)
    nn = np.asarray(nn_distances, dtype=np.float64)
    bad = np.isnan(nn) | np.isinf(nn) | (nn <= 0)
    local_min = float(nn[~bad].min()) if (~bad).any() else np.inf
    stats = comm.host.allgather((local_min, int(bad.sum()), int(nn.size)))
    n_bad, n_all = sum(s[1] for s in stats), sum(s[2] for s in stats)
    if n_bad == n_all:
        raise ValueError(
            f"All {n_bad:,} computed nearest neighbor distances (`nn_distances` attribute) contain invalid "
            "values. Please check the input data.")
    if n_bad:
        if comm.rank == 0:
            logger.warning(
                f"The computed nearest neighbor distances (`nn_distances` attribute) contain {n_bad:,} invalid "
                "values. Setting invalid distances to the minimum positive value found.")
        nn = np.where(~bad, nn, min(s[0] for s in stats))
    return nn


def validate_k(k, n_samples):
    """reference validation.py:595-612."""
    if not isinstance(k, (int, np.integer)) or k < 1:
        raise ValueError(f"k={k} must be a positive integer.")
    if n_samples < 2:
        raise ValueError(f"At least two samples are required to compute nearest neighbors but got {n_samples}.")
    if k >= n_samples:
        raise ValueError(f"k={k} must be smaller than the number of samples {n_samples}.")


def validate_time_x(x, times=None, n_features=None, cast_scalar=False):
    """reference validation.py:23-102: append `times` as the last column of x."""
    x = validate_array(x, "x", ndim=2)
    if cast_scalar and times is not None and (np.isscalar(times) or all(s == 1 for s in np.shape(times))):
        times = np.full(x.shape[0], float(np.asarray(times).reshape(-1)[0]))
    times = validate_array(times, "times", optional=True, ndim=(1, 2))
    if times is not None:
        if times.ndim == 1:
            times = times.reshape(-1, 1)
        elif times.shape[1] != 1:
            raise ValueError("'times' must be a 1D array or a 2D array with 1 column.")
        if x.shape[0] != times.shape[0]:
            raise ValueError(
                "'x' and 'times' must have the same number of samples. "
                f"Got {x.shape[0]} for 'x' and {times.shape[0]} for 'times'.")
        x = np.concatenate((x, times), axis=1)
    if n_features is not None:
        if x.shape[1] == n_features - 1 and times is None:
            raise ValueError(
                f"Expected {n_features} features including 'times' in 'x' but "
                f"only found {x.shape[1]} features and 'times' is not provided.")
        if x.shape[1] != n_features:
            raise ValueError(f"Wrong number of features in 'x'. Expected {n_features} but got {x.shape[1]}.")
    return x
