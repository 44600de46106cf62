"""Estimator namespace (mellon/model.py)."""
from .density_estimator import DensityEstimator  # noqa: F401
from .function_estimator import FunctionEstimator  # noqa: F401
from .time_sensitive_density_estimator import TimeSensitiveDensityEstimator  # noqa: F401
