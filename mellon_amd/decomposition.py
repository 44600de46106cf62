"""Covariance factors (mellon/decomposition.py), device resident.

`_full_rank` / `_standard_low_rank` of the reference return JAX arrays.  Here both factors live
in HBM behind one `mln_fit` handle and are exposed as array-likes: `FactorL` (n x m, the
cell-sharded factor the optimiser streams) and `FactorLp` (m x m).  `np.asarray(factor)` downloads
(chunked for L); nothing is copied to the host unless asked.
"""
import numpy as np

from . import _lib
from .util import DEFAULT_JITTER, ensure_2d

DEFAULT_RANK = 0.99   # reference decomposition.py:17
DEFAULT_SIGMA = 0


class _DeviceFactor:
    ndim = 2
    dtype = np.dtype("float64")

    def __init__(self, fit):
        self.fit = fit

    def __len__(self):
        return self.shape[0]

    def __repr__(self):
        return f"<{self.__class__.__name__} {self.shape[0]:,} x {self.shape[1]:,} float64 on gfx950>"


class FactorL(_DeviceFactor):
    """L with L L^T ~ K (decomposition.py:174-210); rows are this rank's cells."""

    @property
    def shape(self):
        return (self.fit.n, self.fit.m)

    def __array__(self, dtype=None, copy=None):
        out = self.fit.L()
        return out if dtype is None else out.astype(dtype)

    def dot(self, z):
        """L z (the `transform` without mu, inference.py:66-67)."""
        return self.fit.transform(np.asarray(z, dtype=np.float64), 0.0)

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            start, stop, step = idx.indices(self.fit.n)
            if step == 1:
                return self.fit.L(start, max(stop - start, 0))
        return np.asarray(self)[idx]


class FactorLp(_DeviceFactor):
    """Lp with Lp Lp^T = K(landmarks) + jitter I (decomposition.py:79-123)."""

    @property
    def shape(self):
        return (self.fit.m, self.fit.m)

    def __array__(self, dtype=None, copy=None):
        out = self.fit.Lp()
        return out if dtype is None else out.astype(dtype)

    def __getitem__(self, idx):
        return np.asarray(self)[idx]


def _diag_value(sigma, jitter):
    """sigma^2 floored at jitter (decomposition.py:111-112)."""
    s2 = float(np.square(sigma))
    return jitter if s2 < jitter else s2


def _full_rank(x, cov_func, sigma=DEFAULT_SIGMA, jitter=DEFAULT_JITTER, ctx=None):
    """chol(K(x,x) + max(sigma^2, jitter) I), decomposition.py:79-123.  Raises the reference's
    ValueError when a pivot is non-positive / NaN."""
    ctx = ctx or _lib.default_context()
    x = np.ascontiguousarray(ensure_2d(x), dtype=np.float64)
    fit = ctx.fit_prepare(cov_func.lower(x.shape[1]), x, None, _diag_value(sigma, jitter))
    return FactorLp(fit)


def _standard_low_rank(x, cov_func, xu, Lp=None, sigma=DEFAULT_SIGMA, jitter=DEFAULT_JITTER, ctx=None):
    """L = K(x, xu) Lp^-T, decomposition.py:174-210."""
    ctx = ctx or _lib.default_context()
    if not isinstance(x, _lib.DeviceArray):
        x = np.ascontiguousarray(ensure_2d(x), dtype=np.float64)
    xu = np.ascontiguousarray(ensure_2d(xu), dtype=np.float64)
    if isinstance(Lp, FactorLp):
        Lp = np.asarray(Lp)
    fit = ctx.fit_prepare(cov_func.lower(xu.shape[1]), x, xu, _diag_value(sigma, jitter), Lp=Lp)
    return FactorL(fit)


def _nystroem_unavailable(*args, **kwargs):
    raise NotImplementedError(
        "Nystroem rank reduction (gp_type full_nystroem / sparse_nystroem, decomposition.py:126-171,213-266) "
        "is outside the accelerated path of this build (SURVEY.md S8f rank 4).")


_full_decomposition_low_rank = _nystroem_unavailable
_modified_low_rank = _nystroem_unavailable
