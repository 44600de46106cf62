"""Built-in stationary kernels (mellon/cov.py).  Each class only carries parameters; the
arithmetic of k(x, y) lives in csrc/cov_kernels.hip (`leaf_value`), selected by `_kind`."""
from . import _lib
from .base_cov import Add, Covariance, CovariancePair, Mul, Pow  # noqa: F401  (re-exported like the reference)


class Matern32(Covariance):
    """(1 + sqrt3 r/l) exp(-sqrt3 r/l) -- reference cov.py:6-66."""
    _kind = _lib.K_MATERN32

    def __init__(self, ls=1.0, active_dims=None):
        super().__init__()
        self.ls = ls
        self.active_dims = active_dims


class Matern52(Covariance):
    """(1 + sqrt5 r/l + 5 r^2 / 3 l^2) exp(-sqrt5 r/l) -- reference cov.py:103-161."""
    _kind = _lib.K_MATERN52

    def __init__(self, ls=1.0, active_dims=None):
        super().__init__()
        self.ls = ls
        self.active_dims = active_dims


class ExpQuad(Covariance):
    """exp(-r^2 / 2 l^2) -- reference cov.py:205-259."""
    _kind = _lib.K_EXPQUAD

    def __init__(self, ls=1.0, active_dims=None):
        super().__init__()
        self.ls = ls
        self.active_dims = active_dims


class Exponential(Covariance):
    """exp(-r / 2 l) -- reference cov.py:302-356."""
    _kind = _lib.K_EXPONENTIAL

    def __init__(self, ls=1.0, active_dims=None):
        super().__init__()
        self.ls = ls
        self.active_dims = active_dims


class RatQuad(Covariance):
    """(1 + r^2 / 2 alpha l^2)^-alpha -- reference cov.py:399-457 (ctor order alpha, ls)."""
    _kind = _lib.K_RATQUAD

    def __init__(self, alpha=1.0, ls=1.0, active_dims=None):
        super().__init__()
        self.ls = ls
        self.alpha = alpha
        self.active_dims = active_dims


class Linear(Covariance):
    """x . y / l -- reference cov.py:502-556."""
    _kind = _lib.K_LINEAR

    def __init__(self, ls=1.0, active_dims=None):
        super().__init__()
        self.ls = ls
        self.active_dims = active_dims
