"""Covariance factors (mellon/decomposition.py), device resident.

`_full_rank` / `_standard_low_rank` of the reference return JAX arrays.  Here both factors live
in HBM behind one `mln_fit` handle and are exposed as array-likes: `FactorL` (n x m, the
cell-sharded factor the optimiser streams) and `FactorLp` (m x m).  `np.asarray(factor)` downloads
(chunked for L); nothing is copied to the host unless asked.
"""
import logging

import numpy as np

from . import _lib
from .util import DEFAULT_JITTER, ensure_2d

DEFAULT_RANK = 0.99   # reference decomposition.py:17
DEFAULT_SIGMA = 0
logger = logging.getLogger("mellon")


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


def _select_rank(s, rank):
    """Number p of leading eigenpairs `_eigendecomposition` keeps (decomposition.py:51-76), from the
    ascending eigenvalues s: int rank -> min(rank, #positive); float rank -> position of
    rank * (sum of positive eigenvalues) in their descending cumulative sum, at least 1."""
    s = np.asarray(s, dtype=np.float64)
    if np.any(s <= 0):
        logger.warning("Singularity detected in covariance matrix (non-positive eigenvalues). "
                       "This can complicate prediction. Consider raising the jitter.")
    p = int(np.count_nonzero(s > 0))
    if p == 0:
        raise ValueError("The covariance matrix has no positive eigenvalue; increase the jitter.")
    summed = np.cumsum(s[: -p - 1: -1])
    if isinstance(rank, float):
        p = int(np.searchsorted(summed, summed[-1] * rank))
        if p == 0:
            logger.warning(f"Low variance percentage {rank:%} indicated rank=0. Bumping rank to 1.")
            p = 1
    else:
        p = min(int(rank), p)
    if (isinstance(rank, float) and rank < 1) or rank < len(summed):
        frac = summed[min(p, len(summed) - 1)] / summed[-1]
        logger.info(f"Recovering {frac:%} variance in eigendecomposition.")
    return p


def _eigendecomposition(A, rank=DEFAULT_RANK, ctx=None):
    """Top eigenpairs (s, v) of the symmetric A (decomposition.py:23-76); the eigensolver is the
    block-Jacobi kernel behind mln_eigh."""
    ctx = ctx or _lib.default_context()
    s, v = ctx.eigh(np.asarray(A, dtype=np.float64))
    p = _select_rank(s, rank)
    return s[-p:], v[:, -p:]


def _full_decomposition_low_rank(x, cov_func, rank=DEFAULT_RANK, sigma=DEFAULT_SIGMA, jitter=DEFAULT_JITTER,
                                 ctx=None):
    """L = v sqrt(s) from the leading eigenpairs of K(x,x) + max(sigma^2, jitter) I
    (decomposition.py:126-171).  Column signs are arbitrary, as with any eigensolver."""
    ctx = ctx or _lib.default_context()
    x = np.ascontiguousarray(ensure_2d(x), dtype=np.float64)
    W = ctx.kernel_matrix(cov_func.lower(x.shape[1]), x, x)
    W[np.diag_indices_from(W)] += _diag_value(sigma, jitter)
    s, v = _eigendecomposition(W, rank=rank, ctx=ctx)
    fit = _lib.Fit.from_L(ctx, v * np.sqrt(s))
    return FactorL(fit)


def _modified_low_rank(x, cov_func, xu, rank=DEFAULT_RANK, sigma=DEFAULT_SIGMA, jitter=DEFAULT_JITTER, ctx=None):
    """Improved Nystroem factor (decomposition.py:213-266).

    The reference forms QR(C), eigh(W), T = R v and eigh(T / s T^T) = eigh(R W^-1 R^T), L = Q V sqrt(S).
    With W = Lp Lp^T and B = C Lp^-T (the standard low-rank factor), R W^-1 R^T and B^T B share their
    non-zero spectrum S and  Q V sqrt(S) = B U  for the matching eigenvectors U of B^T B, so the same L
    (up to column signs) is  B U[:, -p:]:  one Gram, one m x m eigensolve, one GEMM -- no n x m QR."""
    ctx = ctx or _lib.default_context()
    if not isinstance(x, _lib.DeviceArray):
        x = np.ascontiguousarray(ensure_2d(x), dtype=np.float64)
    xu = np.ascontiguousarray(ensure_2d(xu), dtype=np.float64)
    base = ctx.fit_prepare(cov_func.lower(xu.shape[1]), x, xu, _diag_value(sigma, jitter))
    try:
        S = base.gram_eigh()
        fit = base.project(_select_rank(S, rank))
    finally:
        base.close()
    return FactorL(fit)
