"""mellon_amd -- MI355X-native sparse-GP density core behind Mellon's estimator API.

Mirrors the public namespace of the reference (mellon/__init__.py:14-19): estimators,
`cov`, `parameters`, `inference`, `conditional`, `util`.  All dense arithmetic runs in
libmellon_hip.so (HIP, gfx950, fp64); this package is NumPy + ctypes only.
"""
__version__ = "0.1.0"

from . import util  # noqa: F401
from . import cov  # noqa: F401
from .base_cov import Covariance  # noqa: F401
from .util import GaussianProcessType  # noqa: F401

_LAZY = {
    "DensityEstimator": ".density_estimator",
    "FunctionEstimator": ".function_estimator",
    "TimeSensitiveDensityEstimator": ".time_sensitive_density_estimator",
    "Predictor": ".base_predictor",
    "parameters": None,
    "inference": None,
    "conditional": None,
    "validation": None,
}


def __getattr__(name):
    import importlib
    if name in _LAZY:
        if _LAZY[name] is None:
            return importlib.import_module("." + name, __name__)
        return getattr(importlib.import_module(_LAZY[name], __name__), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
