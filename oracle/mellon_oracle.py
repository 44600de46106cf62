"""TEST INFRASTRUCTURE ONLY -- float64 NumPy/SciPy restatement of Mellon's
sparse-GP density path (reference: settylab/Mellon v1.7.1, pure Python on JAX).

Why a restatement: the reference cannot be imported here (jax, jaxlib, jaxopt,
pynndescent are absent and there is no network), and nothing under
/root/reference may travel to the GPU box.  Every function below cites the
reference file:line whose arithmetic it follows.  Third-party arithmetic on the
path is mapped as follows (all unpinned in the reference's pyproject.toml:24-27):
  * jax.numpy.linalg.cholesky / jax.scipy.linalg.solve_triangular -> scipy.linalg
    (LAPACK potrf/trtrs both sides);
  * jaxopt.ScipyMinimize(method="L-BFGS-B") -> scipy.optimize.minimize with the
    jaxopt defaults (maxiter=500, tol=None => SciPy ftol=2.22e-9, gtol=1e-5,
    maxcor=10) and the analytic gradient of the loss (the reference uses
    autodiff of the same function);
  * sklearn.linear_model.Ridge(alpha=1, fit_intercept=False) -> closed form
    (L^T L + I)^-1 L^T t (checked against sklearn in tests/test_oracle.py);
  * pynndescent.NNDescent (approximate) -> exact Euclidean 1-NN (KD-tree).

PARITY PINNING.  The reference holds exactly one set of absolute golden numbers
on this path: tests/test_reference_results.py:26-63,93-130 (FunctionEstimator,
n=50: predictions, leverage and smoothed observation variance, full and sparse).
tests/test_oracle_golden.py reproduces all six arrays with this file plus
oracle/jax_prng.py (distance, Matern52, ls heuristic, full Cholesky solve, the
sparse `_sparse_solve`, both hat-matrix diagonals and the HC3 variance GP are
pinned that way).  The per-feature ("per-gene") sigma branches are pinned by the
reference's own property -- one fit with a sigma per output equals the scalar
fits column by column, tests/test_pergene_sigma.py -- re-expressed in
tests/test_oracle.py on top of the golden-pinned scalar branches.  The DensityEstimator
log-density itself has no absolute golden vector anywhere in the reference
("parity unpinned" for that output): it is anchored by the pinned building
blocks above, by the reference's analytic/property tests re-expressed in
tests/test_oracle.py, and by strict convexity of the MAP objective.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (mellon_amd) never does.
"""
import math
from collections import namedtuple

import numpy as np
from scipy.linalg import cholesky as _sp_cholesky
from scipy.linalg import solve_triangular as _sp_trsolve
from scipy.optimize import minimize as _sp_minimize
from scipy.special import gammaln

DEFAULT_JITTER = 1e-6          # util.py:48
DEFAULT_RANK = 0.99            # decomposition.py:17
DEFAULT_N_LANDMARKS = 5000     # parameters.py:53
DEFAULT_RANDOM_SEED = 42       # parameters.py:54


# --------------------------------------------------------------------------
# util.py
# --------------------------------------------------------------------------
def ensure_2d(x):
    """util.py:135-147 -- 1-D input becomes n x 1."""
    x = np.asarray(x, dtype=np.float64)
    return np.atleast_2d(x.T).T


def select_active_dims(x, active_dims):
    """util.py:150-171."""
    if active_dims is None:
        return x
    if np.isscalar(active_dims):
        active_dims = [active_dims]
    return x[..., active_dims]


def distance(x, y):
    """util.py:351-366 -- note the +1e-12 INSIDE the sqrt and the clamp."""
    xx = np.sum(x * x, axis=1)[:, None]
    yy = np.sum(y * y, axis=1)[None, :]
    xy = x @ y.T
    sq = xx - 2.0 * xy + yy + 1e-12
    return np.sqrt(np.maximum(sq, 0.0))


def distance_grad(x, eps=1e-12):
    """util.py:369-425 -- y -> (distance (n, m), gradient of the distance w.r.t. y (n, m, d))."""
    xx = np.sum(x * x, axis=1)[:, None]

    def grad(y):
        yy = np.sum(y * y, axis=1)[None, :]
        sq = xx - 2 * np.tensordot(x, y, axes=(1, 1)) + yy + eps
        dist = np.sqrt(np.maximum(sq, 0))
        delta = y[None, :] - x[:, None]
        return dist, delta / (dist[..., None] + eps)

    return grad


def expand_to_inactive(values, target_shape, active_dims):
    """util.py:174-203 -- scatter gradient values into the active columns, zeros elsewhere."""
    if active_dims is None:
        return values
    if np.isscalar(active_dims):
        active_dims = [active_dims]
    full = np.zeros(target_shape, dtype=values.dtype)
    full[..., active_dims] = values
    return full


def stabilize(A, jitter=DEFAULT_JITTER):
    """util.py:269-293."""
    return A + np.eye(A.shape[0]) * jitter


def add_variance(K, M=None, jitter=DEFAULT_JITTER):
    """util.py:296-331."""
    if M is None:
        return stabilize(K, jitter)
    if np.isscalar(M):
        return K + np.eye(K.shape[0]) * max(jitter, M ** 2)
    noise = M @ M.T
    dn = np.diag(noise)
    diff = np.where(dn < jitter, jitter - dn, 0.0)
    return K + noise + np.diag(diff)


def mle(nn_distances, d):
    """util.py:334-348."""
    return gammaln(d / 2 + 1) - (d / 2) * np.log(np.pi) - d * np.log(nn_distances)


# --------------------------------------------------------------------------
# cov.py / base_cov.py -- the kernel-plugin surface
# --------------------------------------------------------------------------
class Covariance:
    """base_cov.py:17-224 (k, __call__, + * **, active_dims, dict round trip)."""

    def __call__(self, x, y):
        return self.k(x, y)

    def __add__(self, other):
        return Add(self, other)

    __radd__ = __add__

    def __mul__(self, other):
        return Mul(self, other)

    __rmul__ = __mul__

    def __pow__(self, other):
        return Pow(self, other)

    def diag(self, x):
        """base_cov.py:71-93 -- k(x_i, x_i) for every row."""
        x = np.asarray(x, dtype=np.float64)
        return np.array([self.k(x[i:i + 1], x[i:i + 1])[0, 0] for i in range(x.shape[0])])

    @staticmethod
    def from_dict(state):
        """base_cov.py:197-224, 272-298 -- accepts the reference's wire format."""
        if not isinstance(state, dict) or state.get("type") != "mellon.Covariance":
            raise ValueError("The passed dict does not seem to define a covariance kernel.")
        cls = _CLASSES[state["metadata"]["classname"]]
        obj = cls.__new__(cls)
        if issubclass(cls, _Pair):
            obj.left = Covariance.from_dict(state["left_data"])
            rd = state["right_data"]
            if isinstance(rd, dict) and rd.get("type") == "mellon.Covariance":
                obj.right = Covariance.from_dict(rd)
            else:
                obj.right = _deserialize(rd)
            obj.active_dims = _deserialize(state.get("active_dims"))
        else:
            for name, val in state["data"].items():
                setattr(obj, name, _deserialize(val))
        return obj


def _serialize(v):
    """util.py:69-92: arrays -> {"type": "jax.numpy", "data": nested lists}, slices, dicts and sets tagged, NumPy
    scalars -> Python scalars, None -> the string "None"."""
    if isinstance(v, np.ndarray):
        return {"type": "jax.numpy", "data": v.tolist()}
    if isinstance(v, np.integer):
        return int(v)
    if isinstance(v, np.floating):
        return float(v)
    if isinstance(v, slice):
        return {"type": "slice", "data": ["None" if q is None else q for q in (v.start, v.stop, v.step)]}
    if isinstance(v, dict):
        return {"type": "dict", "data": {k: _serialize(q) for k, q in v.items()}}
    if isinstance(v, set):
        return {"type": "set", "data": [_serialize(q) for q in v]}
    return "None" if v is None else v


def _metadata(classname, module_name):
    import datetime
    import sys
    return {"classname": classname, "module_name": module_name, "module_version": "1.7.1",
            "serialization_date": datetime.datetime.now().isoformat(), "python_version": sys.version}


def covariance_to_dict(c):
    """base_cov.py:122-160 (leaves: every attribute under "data") and :244-272 (Add / Mul / Pow: left_data,
    right_data -- a covariance state or a serialised scalar -- and active_dims)."""
    if isinstance(c, _Pair):
        right = covariance_to_dict(c.right) if callable(c.right) else _serialize(c.right)
        return {"type": "mellon.Covariance", "left_data": covariance_to_dict(c.left), "right_data": right,
                "active_dims": _serialize(getattr(c, "active_dims", None)),
                "metadata": _metadata(type(c).__name__, "mellon")}
    return {"type": "mellon.Covariance", "data": {k: _serialize(v) for k, v in c.__dict__.items()},
            "metadata": _metadata(type(c).__name__, "mellon.cov")}


def _deserialize(v):
    """util.py:95-132: tagged dicts for arrays / slices, the string "None" for None."""
    if isinstance(v, dict):
        if v["type"] == "jax.numpy":
            return np.array(v["data"])
        if v["type"] == "slice":
            return slice(*[None if (isinstance(q, str) and q == "None") else q for q in v["data"]])
        if v["type"] == "dict":
            return {k: _deserialize(q) for k, q in v["data"].items()}
        if v["type"] == "set":
            return {_deserialize(q) for q in v["data"]}
    if isinstance(v, str) and v == "None":
        return None
    return v


class _Stationary(Covariance):
    def __init__(self, ls=1.0, active_dims=None):
        self.ls = ls
        self.active_dims = active_dims

    def _dist(self, x, y):
        x = select_active_dims(x, self.active_dims)
        y = select_active_dims(y, self.active_dims)
        return distance(x, y)

    def k_grad(self, x):
        """The common frame of every stationary k_grad (e.g. cov.py:83-100): select dims, distance_grad,
        radial rule `_radial_grad(dist[..., None], dgrad)`, expand to the inactive dims."""
        x_shape = x.shape
        xs = select_active_dims(x, self.active_dims)
        dg = distance_grad(xs)

        def k_grad(y):
            y_shape = y.shape
            dist, grad = dg(select_active_dims(y, self.active_dims))
            return expand_to_inactive(self._radial_grad(dist[..., None], grad), x_shape[:-1] + y_shape,
                                      self.active_dims)

        return k_grad


class Matern32(_Stationary):
    def k(self, x, y):
        """cov.py:62-66."""
        r = np.sqrt(3.0) * self._dist(x, y) / self.ls
        return (r + 1) * np.exp(-r)

    def _radial_grad(self, dist, grad):
        """cov.py:86-95."""
        factor = np.sqrt(3.0) / self.ls
        r = -factor * dist
        return r * (factor * grad) * np.exp(r)


class Matern52(_Stationary):
    def k(self, x, y):
        """cov.py:157-161."""
        r = np.sqrt(5.0) * self._dist(x, y) / self.ls
        return (r + np.square(r) / 3 + 1) * np.exp(-r)

    def _radial_grad(self, dist, grad):
        """cov.py:188-197."""
        factor = np.sqrt(5.0) / self.ls
        r = factor * dist
        return -1 / 3 * np.exp(-r) * r * (r + 1) * (factor * grad)


class ExpQuad(_Stationary):
    def k(self, x, y):
        """cov.py:255-259."""
        r = self._dist(x, y) / self.ls
        return np.exp(-np.square(r) / 2)

    def _radial_grad(self, dist, grad):
        """cov.py:288-294."""
        r = dist / self.ls
        return -r * (grad / self.ls) * np.exp(-np.square(r) / 2)


class Exponential(_Stationary):
    def k(self, x, y):
        """cov.py:352-356 (note the non-standard /2)."""
        r = self._dist(x, y) / self.ls
        return np.exp(-r / 2)

    def _radial_grad(self, dist, grad):
        """cov.py:385-391."""
        r = dist / self.ls
        return -1 / 2 * (grad / self.ls) * np.exp(-r / 2)


class RatQuad(_Stationary):
    def __init__(self, alpha=1.0, ls=1.0, active_dims=None):
        """cov.py:428 -- argument order (alpha, ls, active_dims)."""
        self.alpha = alpha
        self.ls = ls
        self.active_dims = active_dims

    def k(self, x, y):
        """cov.py:453-457."""
        r = self._dist(x, y) / self.ls
        return (np.square(r) / (2 * self.alpha) + 1) ** -self.alpha

    def _radial_grad(self, dist, grad):
        """cov.py:486-494."""
        r = dist / self.ls
        return -r * (grad / self.ls) * (np.square(r) / (2 * self.alpha) + 1) ** (-self.alpha - 1)


class Linear(_Stationary):
    def k(self, x, y):
        """cov.py:551-556."""
        x = select_active_dims(x, self.active_dims)
        y = select_active_dims(y, self.active_dims)
        return (x @ y.T) / self.ls

    def k_grad(self, x):
        """cov.py:558-596 -- d (x.y / ls) / dy = x / ls for every y."""
        x_shape = x.shape
        xs = select_active_dims(x, self.active_dims)

        def k_grad(y):
            ys = select_active_dims(y, self.active_dims)
            g = np.repeat(xs[:, None, :], ys.shape[0], axis=1) / self.ls
            return expand_to_inactive(g, x_shape[:-1] + y.shape, self.active_dims)

        return k_grad


class _Pair(Covariance):
    """base_cov.py:227-298."""

    def __init__(self, left, right, active_dims=None):
        self.left = left
        self.right = right
        self.active_dims = active_dims

    def _sel(self, x, y):
        return select_active_dims(x, self.active_dims), select_active_dims(y, self.active_dims)


class Add(_Pair):
    def k(self, x, y):
        """base_cov.py:309-315."""
        x, y = self._sel(x, y)
        if callable(self.right):
            return self.left(x, y) + self.right(x, y)
        return self.left(x, y) + self.right

    def k_grad(self, x):
        """base_cov.py:317-364."""
        x_shape = x.shape
        xs = select_active_dims(x, self.active_dims)
        lg = self.left.k_grad(xs)
        rg = self.right.k_grad(xs) if callable(self.right) else None

        def k_grad(y):
            ys = select_active_dims(y, self.active_dims)
            g = lg(ys) + (rg(ys) if rg is not None else 0.0)
            return expand_to_inactive(g, x_shape[:-1] + y.shape, self.active_dims)

        return k_grad


class Mul(_Pair):
    def k(self, x, y):
        """base_cov.py:375-381."""
        x, y = self._sel(x, y)
        if callable(self.right):
            return self.left(x, y) * self.right(x, y)
        return self.left(x, y) * self.right

    def k_grad(self, x):
        """base_cov.py:383-438 -- product rule (scalar right: scaled left gradient)."""
        x_shape = x.shape
        xs = select_active_dims(x, self.active_dims)
        lg = self.left.k_grad(xs)
        rg = self.right.k_grad(xs) if callable(self.right) else None

        def k_grad(y):
            ys = select_active_dims(y, self.active_dims)
            if rg is None:
                g = lg(ys) * self.right
            else:
                g = lg(ys) * self.right.k(xs, ys)[..., None] + self.left.k(xs, ys)[..., None] * rg(ys)
            return expand_to_inactive(g, x_shape[:-1] + y.shape, self.active_dims)

        return k_grad


class Pow(_Pair):
    def k(self, x, y):
        """base_cov.py:449-453."""
        x, y = self._sel(x, y)
        return self.left(x, y) ** self.right

    def k_grad(self, x):
        """base_cov.py:455-497 -- n k^(n-1) k'."""
        x_shape = x.shape
        xs = select_active_dims(x, self.active_dims)
        bg = self.left.k_grad(xs)

        def k_grad(y):
            ys = select_active_dims(y, self.active_dims)
            g = self.right * (self.left.k(xs, ys)[..., None] ** (self.right - 1)) * bg(ys)
            return expand_to_inactive(g, x_shape[:-1] + y.shape, self.active_dims)

        return k_grad


_CLASSES = {c.__name__: c for c in
            (Matern32, Matern52, ExpQuad, Exponential, RatQuad, Linear, Add, Mul, Pow)}


def compute_cov_func(cov_func_curry, ls, ls_time=None):
    """parameters.py:616-645."""
    if ls_time is not None:
        return cov_func_curry(ls=ls, active_dims=slice(None, -1)) * cov_func_curry(
            ls=ls_time, active_dims=-1)
    return cov_func_curry(ls=ls)


# --------------------------------------------------------------------------
# decomposition.py
# --------------------------------------------------------------------------
_NOT_PD = ("Covariance not positively definite with jitter={jitter}. "
           "Consider increasing the jitter for numerical stabilization.")


def _chol_lower(W, jitter):
    """jnp.linalg.cholesky returns NaNs on failure and the reference turns any
    NaN into ValueError (decomposition.py:115-122); LAPACK raises instead."""
    try:
        L = _sp_cholesky(W, lower=True, check_finite=False)
    except np.linalg.LinAlgError:
        raise ValueError(_NOT_PD.format(jitter=jitter))
    if np.any(np.isnan(L)):
        raise ValueError(_NOT_PD.format(jitter=jitter))
    return L


def full_rank(x, cov_func, sigma=0.0, jitter=DEFAULT_JITTER):
    """decomposition.py:79-123 -- chol(K(x,x) + max(sigma^2, jitter) I)."""
    sigma2 = np.square(sigma)
    sigma2 = np.where(sigma2 < jitter, jitter, sigma2)
    W = stabilize(cov_func(x, x), sigma2)
    return _chol_lower(W, jitter)


def standard_low_rank(x, cov_func, xu, Lp=None, sigma=0.0, jitter=DEFAULT_JITTER):
    """decomposition.py:174-210 -- L = K(x,xu) Lp^-T."""
    C = cov_func(x, xu)
    if Lp is None:
        Lp = full_rank(xu, cov_func, sigma=sigma, jitter=jitter)
    return _sp_trsolve(Lp, C.T, lower=True, check_finite=False).T


def eigendecomposition(A, rank=DEFAULT_RANK):
    """decomposition.py:23-76 -- top eigenpairs of A.  int rank: min(rank, #positive);
    float rank: searchsorted of rank * (sum of positive eigenvalues) in their descending
    cumulative sum (at least 1)."""
    s, v = np.linalg.eigh(A)
    p = int(np.count_nonzero(s > 0))
    summed = np.cumsum(s[: -p - 1: -1])
    if isinstance(rank, float):
        target = summed[-1] * rank
        p = int(np.searchsorted(summed, target))
        if p == 0:
            p = 1
    else:
        p = min(int(rank), p)
    return s[-p:], v[:, -p:]


def full_decomposition_low_rank(x, cov_func, rank=DEFAULT_RANK, sigma=0.0, jitter=DEFAULT_JITTER):
    """decomposition.py:126-171 -- L = v sqrt(s) from the eigenpairs of K(x,x) + sigma2 I."""
    sigma2 = np.square(sigma)
    sigma2 = np.where(sigma2 < jitter, jitter, sigma2)
    W = stabilize(cov_func(x, x), sigma2)
    s, v = eigendecomposition(W, rank=rank)
    return v * np.sqrt(s)


def modified_low_rank(x, cov_func, xu, rank=DEFAULT_RANK, sigma=0.0, jitter=DEFAULT_JITTER):
    """decomposition.py:213-266 -- improved Nystroem: QR of C, eigh of W, eigh of R W^-1 R^T."""
    sigma2 = np.square(sigma)
    sigma2 = np.where(sigma2 < jitter, jitter, sigma2)
    W = stabilize(cov_func(xu, xu), sigma2)
    C = cov_func(x, xu)
    Q, R = np.linalg.qr(C, mode="reduced")
    s, v = eigendecomposition(W, rank=xu.shape[0])
    T = R @ v
    S, V = eigendecomposition(T / s @ T.T, rank=rank)
    return Q @ V * np.sqrt(S)


# --------------------------------------------------------------------------
# parameters.py -- heuristics and decision tables
# --------------------------------------------------------------------------
FULL, FULL_NYSTROEM, SPARSE_CHOLESKY, SPARSE_NYSTROEM = (
    "full", "full_nystroem", "sparse_cholesky", "sparse_nystroem")


def compute_rank(gp_type):
    """parameters.py:88-115."""
    if gp_type in (FULL_NYSTROEM, SPARSE_NYSTROEM):
        return DEFAULT_RANK
    return 1.0


def compute_n_landmarks(gp_type, n_samples, landmarks):
    """parameters.py:118-172."""
    if landmarks is not None:
        return landmarks.shape[0]
    if gp_type in (FULL, FULL_NYSTROEM):
        return n_samples
    if gp_type in (SPARSE_CHOLESKY, SPARSE_NYSTROEM):
        return DEFAULT_N_LANDMARKS
    return min(n_samples, DEFAULT_N_LANDMARKS)


def _rank_is_full(rank, bound):
    return (rank is None
            or (isinstance(rank, (int, np.integer)) and not isinstance(rank, bool) and rank >= bound)
            or (isinstance(rank, float) and rank >= 1.0)
            or rank == 0)


def compute_gp_type(n_landmarks, rank, n_samples):
    """parameters.py:175-240, pinned by tests/test_parameters.py:271-290."""
    if n_landmarks == 0 or n_landmarks >= n_samples:
        return FULL if _rank_is_full(rank, n_samples) else FULL_NYSTROEM
    return SPARSE_CHOLESKY if _rank_is_full(rank, n_landmarks) else SPARSE_NYSTROEM


def compute_landmarks(x, gp_type=None, n_landmarks=DEFAULT_N_LANDMARKS,
                      random_state=DEFAULT_RANDOM_SEED):
    """parameters.py:243-291 -- sklearn k_means centroids (third party)."""
    if n_landmarks == 0:
        return None
    x = ensure_2d(x)
    if n_landmarks >= x.shape[0]:
        return None
    from sklearn.cluster import k_means
    return k_means(x, n_landmarks, n_init=1, random_state=random_state)[0]


def exact_nn_distances(x):
    """Stand-in for parameters.py:352-433 (pynndescent k=1, approximate):
    the exact Euclidean nearest-neighbour distance of every row."""
    from sklearn.neighbors import KDTree, BallTree
    x = ensure_2d(x)
    tree = (KDTree if x.shape[1] <= 20 else BallTree)(x)
    dist, _ = tree.query(x, k=2)
    return dist[:, 1]


def validate_nn_distances(nn):
    """validation.py:528-592."""
    nn = np.asarray(nn, dtype=np.float64)
    bad = np.isnan(nn) | np.isinf(nn) | (nn <= 0)
    if np.all(bad):
        raise ValueError("All computed nearest neighbor distances contain invalid values.")
    return np.where(~bad, nn, np.min(nn[~bad]))


def compute_mu(nn_distances, d):
    """parameters.py:586-599 -- jnp.quantile default == np.quantile 'linear'."""
    return float(np.quantile(mle(nn_distances, d), 0.01)) - 10


def compute_ls(nn_distances):
    """parameters.py:602-613."""
    return float(np.exp(np.log(nn_distances).mean() + 3.0))


def compute_initial_value(nn_distances, d, mu, L):
    """parameters.py:877-896 -- Ridge(alpha=1, no intercept) closed form."""
    target = mle(nn_distances, d) - mu
    G = L.T @ L + np.eye(L.shape[1])
    c = _sp_cholesky(G, lower=True, check_finite=False)
    rhs = L.T @ target
    return _sp_trsolve(c.T, _sp_trsolve(c, rhs, lower=True), lower=False)


# --------------------------------------------------------------------------
# inference.py
# --------------------------------------------------------------------------
def nn_likelihood_constants(r, d):
    """inference.py:83-85 -- V and Vdr of the nearest-neighbour likelihood."""
    const = (d * np.log(np.pi) / 2) - gammaln(d / 2 + 1)
    V = np.log(r) * d + const
    Vdr = np.log(d) + ((d - 1) * np.log(r)) + const
    return V, Vdr


def loss_and_grad(z, L, mu, V, Vdr):
    """inference.py:35-92,167-192: loss(z) = -(prior(z) + lik(L z + mu));
    gradient derived analytically: z + L^T (exp(f+V) - 1)."""
    k = z.shape[0]
    f = L @ z + mu
    A = np.exp(f + V)
    loss = 0.5 * np.dot(z, z) + (k / 2) * np.log(2 * np.pi) - np.sum((f + Vdr) - A)
    grad = z + L.T @ (A - 1.0)
    return loss, grad


LBFGSB_OPTIONS = dict(maxiter=500)   # jaxopt.ScipyMinimize default; SciPy fills the rest
# The reference's default stopping rule (ftol 2.2e-9) leaves the log-density ~5e-5 (relative) away
# from the unique optimum of the strictly convex objective, and a 1e-15 relative perturbation of L
# moves its answer by the same amount (tests/test_oracle.py::test_default_tolerance_floor): the
# reference output is only reproducible to ~1e-4 across BLAS roundings.  Parity at 1e-5 is therefore
# defined against the optimum itself, reached with these options (same SciPy routine).
LBFGSB_TIGHT = dict(maxiter=20000, maxfun=200000, maxcor=30, ftol=1e-15, gtol=1e-9)

MapResult = namedtuple("MapResult", "pre_transformation loss n_eval n_iter status")


def minimize_lbfgsb(fun_and_grad, z0, options=None):
    """inference.py:272-288 via jaxopt.ScipyMinimize(method='L-BFGS-B')."""
    opts = dict(LBFGSB_OPTIONS)
    if options:
        opts.update(options)
    res = _sp_minimize(fun_and_grad, np.asarray(z0, dtype=np.float64), jac=True,
                       method="L-BFGS-B", options=opts)
    return MapResult(res.x, float(res.fun), int(res.nfev), int(res.nit), int(res.status))


def minimize_adam(fun_and_grad, z0, n_iter=100, init_learn_rate=1e-1):    # inference.py:28-29 defaults
    """inference.py:222-269 with jax.example_libraries.optimizers.adam restated (b1 = 0.9, b2 = 0.999, eps = 1e-8,
    bias-corrected moments; step size exp(-0.01 i) * init_learn_rate).  JAX is absent here, so the update rule is the
    published one, not checked against the library: parity for optimizer="adam" is pinned only through the
    reference's own property (tests/test_density_estimator.py:66-74: within 2e-3 of the L-BFGS-B density)."""
    z = np.array(z0, dtype=np.float64)
    m1, m2 = np.zeros_like(z), np.zeros_like(z)
    losses = []
    for i in range(n_iter):
        value, g = fun_and_grad(z)
        losses.append(value)
        m1 = 0.1 * g + 0.9 * m1
        m2 = 0.001 * np.square(g) + 0.999 * m2
        z = z - np.exp(-1e-2 * i) * init_learn_rate * (m1 / (1 - 0.9 ** (i + 1))) / (np.sqrt(m2 / (1 - 0.999 ** (i + 1))) + 1e-8)
    return z, np.asarray(losses)


def laplace_std(z, L, mu, V):
    """inference.py:291-338 -- diag of the Hessian in closed form:
    1 + sum_i L_ij^2 exp(f_i+V_i), clipped at 1e-8, std = 1/sqrt."""
    a = np.exp(L @ z + mu + V)
    h = 1.0 + (L * L).T @ a
    return 1.0 / np.sqrt(np.maximum(h, 1e-8))


# --------------------------------------------------------------------------
# conditional.py -- predictors (mean only)
# --------------------------------------------------------------------------
def _get_L(x, cov_func, jitter=DEFAULT_JITTER, y_cov_factor=None, K=None):
    """conditional.py:69-81."""
    if K is None:
        K = cov_func(x, x)
    return _chol_lower(add_variance(K, y_cov_factor, jitter=jitter), jitter)


def sigma_to_y_cov_factor(sigma, y_cov_factor, n):
    """conditional.py:100-135."""
    if sigma is None and y_cov_factor is None:
        raise ValueError("No input uncertainty specified.")
    if y_cov_factor is not None and sigma is not None and np.any(np.asarray(sigma) > 0):
        raise ValueError("One can specify either `sigma` or `y_cov_factor`, not both.")
    if y_cov_factor is not None:
        return y_cov_factor
    sigma = np.asarray(sigma, dtype=np.float64)
    if sigma.ndim == 0:
        return np.eye(n) * sigma
    if sigma.ndim == 1:
        return np.diag(sigma)
    out = np.zeros((n,) + sigma.shape)                  # conditional.py:122-131: a leading dimension for the diagonal
    for i in range(n):
        out[i, i, ...] = sigma[i]
    return out


def is_per_feature_sigma(sigma, y):
    """conditional.py:13-36: sigma of shape (p,), (1, p) or (n, p) against a 2-D y of shape (n, p)."""
    if sigma is None or np.ndim(sigma) == 0:
        return False
    sigma, y = np.asarray(sigma), np.asarray(y)
    if sigma.ndim == 2 and sigma.shape[0] == 1 and y.ndim == 2 and sigma.shape[1] == y.shape[1]:
        return True
    if sigma.ndim == 2 and y.ndim == 2 and sigma.shape == y.shape:
        return True
    return bool(sigma.ndim == 1 and y.ndim == 2 and sigma.shape[0] == y.shape[1])


def normalize_per_feature_sigma(sigma):
    """conditional.py:39-43: (1, p) -> (p,)."""
    sigma = np.asarray(sigma, dtype=np.float64)
    return sigma[0] if sigma.ndim == 2 and sigma.shape[0] == 1 else sigma


def _sigma_columns(sigma_pf, p):
    """The per-column noise the reference's vmap hands to each solve: sigma_pf[g] for (p,), sigma_pf[:, g] for (n, p)."""
    return [sigma_pf[:, g] if sigma_pf.ndim == 2 else sigma_pf[g] for g in range(p)]


def _full_leverage_one(K, sigma_g, jitter):
    """conditional.py:313-323,385-403: 1 - sigma^2 diag((K + sigma^2 I + jitter I)^-1)."""
    n = K.shape[0]
    Lf = _sp_cholesky(stabilize(K + sigma_g ** 2 * np.eye(n), jitter), lower=True)
    Linv = _sp_trsolve(Lf, np.eye(n), lower=True)
    return 1 - sigma_g ** 2 * np.sum(np.square(Linv), axis=0)


def _landmarks_leverage_one(B, K_uu, sigma_g, jitter):
    """conditional.py:600-605,672-685: diag(B M^-1 B^T), M = sigma^2 K_uu + B^T B + jitter I."""
    M = stabilize(sigma_g ** 2 * K_uu + B.T @ B, jitter)
    return np.sum((B @ np.linalg.inv(M)) * B, axis=1)


def sparse_solve(Lp, A, r_l, A_l):
    """conditional.py:57-66."""
    LBB = stabilize(A_l @ A.T, 1.0)
    L_B = _sp_cholesky(LBB, lower=True, check_finite=False)
    c = _sp_trsolve(L_B, A @ r_l, lower=True)
    w = _sp_trsolve(Lp.T, _sp_trsolve(L_B.T, c, lower=False), lower=False)
    return w, L_B


class Predictor:
    """base_predictor.py:180-257 (mean / __call__ only)."""

    def __init__(self, cov_func, centers, weights, mu, n_obs):
        self.cov_func, self.centers, self.weights = cov_func, centers, weights
        self.mu, self.n_obs = mu, n_obs
        self.n_input_features = centers.shape[1]

    def __call__(self, Xnew, normalize=False):
        Xnew = ensure_2d(Xnew)
        if Xnew.shape[1] != self.n_input_features:
            raise ValueError(
                f"The predictor was trained on data with {self.n_input_features} features "
                f"but the input has {Xnew.shape[1]} features.")
        out = self.mu + self.cov_func(Xnew, self.centers) @ self.weights
        if normalize:
            out = out - math.log(self.n_obs)
        return out

    mean = __call__

    # -- wire format of base_predictor.py:541-595 (only what the mean needs) -----------------------------------------
    def to_dict(self, classname="LandmarksConditionalCholesky", center_name="landmarks", d=None, d_method=None):
        """__getstate__: every name of `_state_variables` plus n_input_features, n_obs, d, d_method and the set itself
        under "data", the covariance state, the metadata block.  center_name: "landmarks" (conditional.py:839-841) or
        "x" (:276-278)."""
        names = {center_name, "weights", "mu", "jitter", "sigma", "per_feature_sigma"}
        vals = {center_name: self.centers, "weights": self.weights, "mu": self.mu, "jitter": DEFAULT_JITTER,
                "sigma": None, "per_feature_sigma": False, "n_input_features": self.n_input_features,
                "n_obs": self.n_obs, "d": d, "d_method": d_method, "_state_variables": names}
        return {"data": {k: _serialize(v) for k, v in vals.items()}, "cov_func": covariance_to_dict(self.cov_func),
                "metadata": _metadata(classname, "mellon.conditional")}

    @staticmethod
    def from_dict(state):
        """__setstate__ (base_predictor.py:583-595): every entry of "data" becomes an attribute; the covariance comes
        from "cov_func".  States written by mellon 1.3.1 carry neither n_obs nor _state_variables
        (tests/test_density_estimator.py:139-151)."""
        data = {k: _deserialize(v) for k, v in state["data"].items()}
        centers = data["landmarks"] if "landmarks" in data else data["x"]
        p = Predictor(Covariance.from_dict(state["cov_func"]), np.asarray(centers, dtype=np.float64),
                      np.asarray(data["weights"], dtype=np.float64), data["mu"], data.get("n_obs"))
        p.d, p.d_method = data.get("d"), data.get("d_method")
        return p

    def gradient(self, Xnew, h=None):
        """base_predictor.py:490-505: d mean / d x per row.  The reference uses jax.jacrev of `_mean`;
        the restatement differentiates the same function by Richardson-extrapolated central differences
        (error ~ h^4 |f^(5)|), independent of any analytic rule -- so it checks the rules."""
        Xnew = ensure_2d(np.asarray(Xnew, dtype=np.float64))
        n, d = Xnew.shape
        if h is None:       # a step well inside the shortest length scale of the kernel (and of the data range)
            def _min_ls(c):
                if hasattr(c, "left"):
                    r = _min_ls(c.right) if callable(c.right) else np.inf
                    return min(_min_ls(c.left), r)
                return float(getattr(c, "ls", np.inf))
            h = 1e-3 * min(np.maximum(np.abs(Xnew).max(), 1.0), _min_ls(self.cov_func))
        out = np.empty((n, d))
        for k in range(d):
            e = np.zeros(d)
            e[k] = h
            d1 = (self(Xnew + e) - self(Xnew - e)) / (2 * h)
            d2 = (self(Xnew + 2 * e) - self(Xnew - 2 * e)) / (4 * h)
            out[:, k] = (4 * d1 - d2) / 3
        return out

    def analytic_gradient(self, Xnew):
        """sum_j w_j d k(x_i, c_j) / d x_i from the covariance's own `k_grad` rules (cov.py:68-596,
        base_cov.py:317-497): k_grad(c)(x)[j, i] is the derivative with respect to x_i."""
        Xnew = ensure_2d(np.asarray(Xnew, dtype=np.float64))
        return np.einsum("j,jik->ik", self.weights, self.cov_func.k_grad(self.centers)(Xnew))

    def hessian(self, Xnew, h=None):
        """base_predictor.py:507-521: jacfwd(jacrev(mean)) per row.  The restatement differentiates the analytic
        gradient above once more by Richardson-extrapolated central differences (error ~ h^4), so the closed-form
        second derivatives of the device path are checked against rules they do not share."""
        Xnew = ensure_2d(np.asarray(Xnew, dtype=np.float64))
        n, d = Xnew.shape
        if h is None:
            def _min_ls(c):
                if hasattr(c, "left"):
                    r = _min_ls(c.right) if callable(c.right) else np.inf
                    return min(_min_ls(c.left), r)
                return float(getattr(c, "ls", np.inf))
            # the differentiated function is analytic and known to ~1e-16, so a small step is affordable: truncation
            # ~ h^4 f^(5) matters near the centres, where Matern32 / Exponential have unbounded higher derivatives
            h = 1e-4 * min(np.maximum(np.abs(Xnew).max(), 1.0), _min_ls(self.cov_func))
        out = np.empty((n, d, d))
        for k in range(d):
            e = np.zeros(d)
            e[k] = h
            d1 = (self.analytic_gradient(Xnew + e) - self.analytic_gradient(Xnew - e)) / (2 * h)
            d2 = (self.analytic_gradient(Xnew + 2 * e) - self.analytic_gradient(Xnew - 2 * e)) / (4 * h)
            out[:, :, k] = (4 * d1 - d2) / 3
        return 0.5 * (out + np.swapaxes(out, 1, 2))

    def hessian_log_determinant(self, Xnew):
        """base_predictor.py:523-539."""
        return np.linalg.slogdet(self.hessian(Xnew))

    # with_uncertainty state: L (factor on the centres) and W = L^-T diag(std)
    L = None
    W = None
    Cs = None
    # obs_variance state (conditional.py:305-362,563-645) and what leverage() needs
    variance_weights = None
    variance_mu = 0.0
    sigma = None
    jitter = DEFAULT_JITTER
    kind = None            # "full" | "landmarks"

    def leverage(self, Xnew):
        """base_predictor.py:263-288 -> conditional._leverage (:373-403 full: the training leverage whatever Xnew;
        :660-685 landmarks: diag(B M^-1 B^T), B = cov(Xnew, xu), M = sigma^2 K_uu + B^T B + jitter I)."""
        Xnew = ensure_2d(Xnew)
        if np.ndim(self.sigma) >= 1:       # per-feature sigma: one column per output (conditional.py:389-397,672-680)
            sig = normalize_per_feature_sigma(self.sigma)
            if self.kind == "full":
                K = self.cov_func(self.centers, self.centers)
                return np.stack([_full_leverage_one(K, sg, self.jitter) for sg in sig], axis=1)
            B = self.cov_func(Xnew, self.centers)
            K_uu = self.L @ self.L.T if self.L is not None else self.cov_func(self.centers, self.centers)
            return np.stack([_landmarks_leverage_one(B, K_uu, sg, self.jitter) for sg in sig], axis=1)
        s2 = float(self.sigma) ** 2
        if self.kind == "full":
            x = self.centers
            n = x.shape[0]
            Lf = _sp_cholesky(stabilize(self.cov_func(x, x) + s2 * np.eye(n), self.jitter), lower=True)
            Linv = _sp_trsolve(Lf, np.eye(n), lower=True)
            return 1 - s2 * np.sum(np.square(Linv), axis=0)
        B = self.cov_func(Xnew, self.centers)
        K_uu = self.L @ self.L.T if self.L is not None else self.cov_func(self.centers, self.centers)
        M = stabilize(s2 * K_uu + B.T @ B, self.jitter)
        return np.sum((B @ np.linalg.inv(M)) * B, axis=1)

    def loo_residuals_squared(self, Xnew, y):
        """base_predictor.py:290-324 -- HC3: r^2 / (1 - h)^2."""
        r = np.asarray(y) - self(Xnew)
        h = self.leverage(Xnew)
        if r.ndim > h.ndim:
            h = h[..., None]
        return r ** 2 / (1 - h) ** 2

    def obs_variance(self, Xnew):
        """base_predictor.py:330-355."""
        if self.variance_weights is None:
            raise ValueError("The predictor was computed without obs_variance. Recompute setting `obs_variance=True`.")
        return self.variance_mu + self.cov_func(ensure_2d(Xnew), self.centers) @ self.variance_weights

    def covariance(self, Xnew, diag=True):
        """conditional.py:409-422,930-945."""
        Xnew = ensure_2d(Xnew)
        Kus = self.cov_func(self.centers, Xnew)
        A = _sp_trsolve(self.L, Kus, lower=True)
        if self.Cs is not None:                               # conditional.py:707-716
            Cc = _sp_trsolve(self.Cs, Kus, lower=True)
            if diag:
                return self.cov_func.diag(Xnew) - np.sum(np.square(A), axis=0) + np.sum(np.square(Cc), axis=0)
            return self.cov_func(Xnew, Xnew) - A.T @ A + Cc.T @ Cc
        if diag:
            return self.cov_func.diag(Xnew) - np.sum(np.square(A), axis=0)
        return self.cov_func(Xnew, Xnew) - A.T @ A

    def mean_covariance(self, Xnew, diag=True):
        """conditional.py:423-440,947-963."""
        cov_L = self.cov_func(ensure_2d(Xnew), self.centers) @ self.W
        return np.sum(cov_L * cov_L, axis=1) if diag else cov_L @ cov_L.T

    def uncertainty(self, Xnew, diag=True):
        """base_predictor.py:390-428."""
        return self.covariance(Xnew, diag) + self.mean_covariance(Xnew, diag)

    def attach_uncertainty(self, Lf, std):
        """conditional.py:853-867 (W = L^-T diag(std)); the full conditional reaches the same W through
        y_cov_factor = L diag(std) (inference.py:357-372, conditional.py:340-362)."""
        self.L = Lf
        self.W = _sp_trsolve(Lf.T, np.diag(np.broadcast_to(std, (Lf.shape[0],))), lower=False)
        return self


def full_conditional(x, y, mu, cov_func, L=None, sigma=0.0, jitter=DEFAULT_JITTER,
                     y_is_mean=False, with_uncertainty=False, obs_variance=False, y_cov_factor=None):
    """conditional.py:183-362, scalar and per-feature sigma, or the caller's own noise factor `y_cov_factor`."""
    x = ensure_2d(x)
    n = x.shape[0]
    per_feature = is_per_feature_sigma(sigma, y)
    if per_feature:                                            # conditional.py:239-251: one solve per output
        K = cov_func(x, x)
        sig_pf = normalize_per_feature_sigma(sigma)
        r = y - mu
        cols = []
        for g, sg in enumerate(_sigma_columns(sig_pf, r.shape[1])):
            Lg = _sp_cholesky(stabilize(K + np.diag(np.broadcast_to(sg ** 2, (n,))), jitter), lower=True)
            cols.append(_sp_trsolve(Lg.T, _sp_trsolve(Lg, r[:, g], lower=True), lower=False))
        w = np.stack(cols, axis=1)
    else:
        if L is None:
            if y_is_mean:
                L = _get_L(x, cov_func, jitter)
            else:
                L = _get_L(x, cov_func, jitter, sigma_to_y_cov_factor(sigma, y_cov_factor, n))
        r = y - mu
        w = _sp_trsolve(L.T, _sp_trsolve(L, r, lower=True), lower=False)
    pred = Predictor(cov_func, x, w, mu, n)
    pred.kind, pred.sigma, pred.jitter = "full", sigma, jitter
    if obs_variance:                                          # conditional.py:305-362
        K = cov_func(x, x)
        resid = y - (mu + K @ w)
        pred.variance_mu = 0.0
        if np.ndim(sigma) >= 1:
            sig_pf = normalize_per_feature_sigma(sigma)
            h = np.stack([_full_leverage_one(K, sg, jitter) for sg in sig_pf], axis=1)
            cr2 = resid ** 2 / (1 - h) ** 2
            cols = []
            for g, sg in enumerate(sig_pf):
                Lv = _sp_cholesky(stabilize(K + sg ** 2 * np.eye(n), jitter), lower=True)
                cols.append(_sp_trsolve(Lv.T, _sp_trsolve(Lv, cr2[:, g], lower=True), lower=False))
            pred.variance_weights = np.stack(cols, axis=1)
        else:
            s2 = float(sigma) ** 2
            Lv = _sp_cholesky(stabilize(K + s2 * np.eye(n), jitter), lower=True)
            Linv = _sp_trsolve(Lv, np.eye(n), lower=True)
            h = 1 - s2 * np.sum(np.square(Linv), axis=0)
            if resid.ndim > h.ndim:
                h = h[..., None]
            cr2 = resid ** 2 / (1 - h) ** 2
            pred.variance_weights = _sp_trsolve(Lv.T, _sp_trsolve(Lv, cr2, lower=True), lower=False)
        pred.corrected_r2 = cr2
    if with_uncertainty and per_feature:                       # conditional.py:288-291: noise-free covariance, no W
        pred.L = _get_L(x, cov_func, jitter)
    elif with_uncertainty:                                     # conditional.py:292-304
        ycf = sigma_to_y_cov_factor(sigma, y_cov_factor, n)
        pred.L = L
        pred.W = _sp_trsolve(L.T, _sp_trsolve(L, ycf, lower=True), lower=False)
    return pred


def landmarks_conditional(x, xu, y, mu, cov_func, Lp=None, sigma=0.0,
                          jitter=DEFAULT_JITTER, y_is_mean=False, with_uncertainty=False, obs_variance=False,
                          y_cov_factor=None):
    """conditional.py:455-645 (scalar, element-wise and per-feature sigma)."""
    x, xu = ensure_2d(x), ensure_2d(xu)
    Kuf = cov_func(xu, x)
    if Lp is None:
        Lp = _get_L(xu, cov_func, jitter)
    A = _sp_trsolve(Lp, Kuf, lower=True)
    r = y - mu
    per_feature = is_per_feature_sigma(sigma, y)
    L_B = None
    if per_feature:                                            # conditional.py:526-545
        sig_pf = normalize_per_feature_sigma(sigma)
        cols = []
        for g, sg in enumerate(_sigma_columns(sig_pf, r.shape[1])):
            s2 = np.square(sg)
            cols.append(sparse_solve(Lp, A, r[:, g] / s2, A / s2)[0])
        w = np.stack(cols, axis=1)
    else:
        if y_is_mean:
            r_l, A_l = r, A
        else:
            sigma2 = np.square(sigma)        # conditional.py:155-159
            r_l, A_l = r / sigma2, A / sigma2
        w, L_B = sparse_solve(Lp, A, r_l, A_l)
    pred = Predictor(cov_func, xu, w, mu, x.shape[0])
    pred.kind, pred.sigma, pred.jitter = "landmarks", sigma, jitter
    if obs_variance:                                          # conditional.py:589-645
        B = Kuf.T
        K_uu = Lp @ Lp.T
        rr = y - (mu + B @ w)
        pred.variance_mu = 0.0
        if np.ndim(sigma) >= 1:
            sig_pf = normalize_per_feature_sigma(sigma)
            h = np.stack([_landmarks_leverage_one(B, K_uu, sg, jitter) for sg in sig_pf], axis=1)
            cr2 = rr ** 2 / (1 - h) ** 2
            pred.variance_weights = np.stack(
                [sparse_solve(Lp, A, cr2[:, g] / sg ** 2, A / sg ** 2)[0] for g, sg in enumerate(sig_pf)], axis=1)
        else:
            s2 = float(sigma) ** 2
            h = _landmarks_leverage_one(B, K_uu, float(sigma), jitter)
            if rr.ndim > h.ndim:
                h = h[..., None]
            cr2 = rr ** 2 / (1 - h) ** 2
            pred.variance_weights, _ = sparse_solve(Lp, A, cr2 / s2, A / s2)
        pred.corrected_r2 = cr2
    if with_uncertainty:                                      # conditional.py:571-577
        pred.L = Lp
        if not per_feature:
            pred.Cs = Lp @ L_B
        if y_is_mean:                                         # conditional.py:579-587
            Cm = _sp_trsolve(L_B, A @ y_cov_factor, lower=True)
            pred.W = _sp_trsolve(Lp.T, _sp_trsolve(L_B.T, Cm, lower=False), lower=False)
    return pred


def landmarks_conditional_cholesky(xu, z, mu, cov_func, n_obs, L=None,
                                   jitter=DEFAULT_JITTER, sigma=0.0, obs_variance=False, obs_x=None, obs_y=None):
    """conditional.py:750-897 -- weights = Lp^-T z; with obs_variance the HC3 residuals of (obs_x, obs_y) feed a
    second landmark GP (:870-897; the leverage there still sees K_uu = cov(xu, xu) because `L` is attached later)."""
    xu = ensure_2d(xu)
    if L is None:
        L = _get_L(xu, cov_func, jitter)
    w = _sp_trsolve(L.T, z, lower=False)
    pred = Predictor(cov_func, xu, w, mu, n_obs)
    pred.kind, pred.sigma, pred.jitter = "landmarks", sigma, jitter
    if obs_variance:
        x = ensure_2d(obs_x)
        B = cov_func(x, xu)
        h = _landmarks_leverage_one(B, cov_func(xu, xu), float(sigma), jitter)
        r = obs_y - (mu + B @ w)
        if r.ndim > h.ndim:
            h = h[..., None]
        cr2 = r ** 2 / (1 - h) ** 2
        A = _sp_trsolve(L, B.T, lower=True)
        s2 = float(sigma) ** 2
        pred.variance_mu = 0.0
        pred.variance_weights, _ = sparse_solve(L, A, cr2 / s2, A / s2)
        pred.corrected_r2 = cr2
    return pred


def compute_conditional(x, landmarks, z, y, mu, cov_func, L, Lp=None, sigma=0.0,
                        jitter=DEFAULT_JITTER, y_is_mean=False, with_uncertainty=False, obs_variance=False):
    """inference.py:375-508 dispatch."""
    if landmarks is None:
        return full_conditional(x, y, mu, cov_func, Lp, sigma=sigma, jitter=jitter,
                                y_is_mean=y_is_mean, with_uncertainty=with_uncertainty, obs_variance=obs_variance)
    if z is not None and z.shape[0] == landmarks.shape[0]:
        return landmarks_conditional_cholesky(landmarks, z, mu, cov_func, x.shape[0], Lp,
                                              jitter=jitter)
    return landmarks_conditional(x, landmarks, y, mu, cov_func, None, sigma=sigma,
                                 jitter=jitter, y_is_mean=y_is_mean, with_uncertainty=with_uncertainty,
                                 obs_variance=obs_variance)


# --------------------------------------------------------------------------
# estimators (density_estimator.py:404-581, function_estimator.py:295-374,
# time_sensitive_density_estimator.py:608-665) as plain functions
# --------------------------------------------------------------------------
DensityFit = namedtuple(
    "DensityFit",
    "log_density_x pre_transformation L Lp landmarks mu ls d nn_distances cov_func "
    "initial_value loss n_eval gp_type predict")


def density_fit(x, cov_func_curry=Matern52, n_landmarks=None, rank=None, landmarks=None,
                nn_distances=None, d=None, mu=None, ls=None, ls_factor=1.0, cov_func=None,
                jitter=DEFAULT_JITTER, initial_value=None, lbfgsb_options=None,
                ls_time=None, random_state=DEFAULT_RANDOM_SEED):
    """DensityEstimator.fit_predict with the attribute pipeline of
    density_estimator.py:404-444 (all four gp types)."""
    x = ensure_2d(x)
    n = x.shape[0]
    if n_landmarks is None:
        n_landmarks = compute_n_landmarks(None, n, landmarks)
    if rank is None:
        rank = compute_rank(None)
    gp_type = compute_gp_type(n_landmarks, rank, n)
    if nn_distances is None:
        nn_distances = exact_nn_distances(x)
    nn_distances = validate_nn_distances(nn_distances)
    if d is None:
        d = x.shape[1] if ls_time is None else x.shape[1] - 1
        if d > 50:
            raise ValueError("The detected dimensionality of the data is over 50")
    if mu is None:
        mu = compute_mu(nn_distances, d)
    if ls is None:
        ls = compute_ls(nn_distances) * ls_factor
    if cov_func is None:
        cov_func = compute_cov_func(cov_func_curry, ls, ls_time)
    if landmarks is None and gp_type in (SPARSE_CHOLESKY, SPARSE_NYSTROEM):
        landmarks = compute_landmarks(x, gp_type, n_landmarks, random_state)
    if gp_type == FULL:
        landmarks = None
        Lp = full_rank(x, cov_func, sigma=0.0, jitter=jitter)
        L = Lp                                              # parameters.py:847-850
    elif gp_type == FULL_NYSTROEM:                          # parameters.py:851-854; Lp stays None (:686-714)
        landmarks = None
        Lp = None
        L = full_decomposition_low_rank(x, cov_func, rank=rank, jitter=jitter)
    elif gp_type == SPARSE_NYSTROEM:                        # parameters.py:866-874
        Lp = None
        L = modified_low_rank(x, cov_func, landmarks, rank=rank, jitter=jitter)
    else:
        Lp = full_rank(landmarks, cov_func, sigma=0.0, jitter=jitter)
        L = standard_low_rank(x, cov_func, landmarks, Lp=Lp)
    if initial_value is None:
        initial_value = compute_initial_value(nn_distances, d, mu, L)
    V, Vdr = nn_likelihood_constants(nn_distances, d)
    res = minimize_lbfgsb(lambda z: loss_and_grad(z, L, mu, V, Vdr), initial_value,
                          lbfgsb_options)
    z = res.pre_transformation
    log_density_x = L @ z + mu                              # inference.py:354
    predict = compute_conditional(x, landmarks, z, log_density_x, mu, cov_func, L, Lp,
                                  sigma=None, jitter=jitter, y_is_mean=True)
    return DensityFit(log_density_x, z, L, Lp, landmarks, mu, ls, d, nn_distances, cov_func,
                      initial_value, res.loss, res.n_eval, gp_type, predict)


def function_fit(x, y, sigma, cov_func_curry=Matern52, n_landmarks=None, landmarks=None,
                 nn_distances=None, mu=0.0, ls=None, ls_factor=1.0, cov_func=None,
                 jitter=DEFAULT_JITTER, y_is_mean=False, random_state=DEFAULT_RANDOM_SEED,
                 with_uncertainty=False, obs_variance=False):
    """FunctionEstimator.fit (function_estimator.py:295-374) -> Predictor."""
    x = ensure_2d(x)
    n = x.shape[0]
    if n_landmarks is None:
        n_landmarks = compute_n_landmarks(None, n, landmarks)
    gp_type = compute_gp_type(n_landmarks, 1.0, n)
    if cov_func is None:
        if ls is None:
            if nn_distances is None:
                nn_distances = exact_nn_distances(x)
            ls = compute_ls(validate_nn_distances(nn_distances)) * ls_factor
        cov_func = compute_cov_func(cov_func_curry, ls)
    if landmarks is None:
        landmarks = compute_landmarks(x, gp_type, n_landmarks, random_state)
    return compute_conditional(x, landmarks, None, np.asarray(y, dtype=np.float64), mu,
                               cov_func, None, None, sigma, jitter=jitter,
                               y_is_mean=y_is_mean, with_uncertainty=with_uncertainty, obs_variance=obs_variance)


def compute_ls_time(nn_distances, x_with_time, cov_func_curry=Matern52, density_fit_kwargs=None):
    """compute_ls_time.py:12-105: a density fit per time point, the correlation matrix of the predicted log-densities
    (each evaluated on ALL states), and the length scale minimising |cov(ls)(dt, 0) - corr|_F over log ls by
    L-BFGS-B from log ls = 0 (jaxopt.ScipyMinimize defaults; central-difference gradient here)."""
    x_with_time = np.asarray(x_with_time, dtype=np.float64)
    times, states = x_with_time[:, -1], x_with_time[:, :-1]
    unique_times = np.unique(times)
    dens = []
    for t in unique_times:
        mask = times == t
        fit = density_fit(states[mask], cov_func_curry=cov_func_curry, nn_distances=nn_distances[mask],
                          **(density_fit_kwargs or {}))
        dens.append(fit.predict(states))
    corrs = np.corrcoef(np.stack(dens))
    nt = len(unique_times)
    delta_t = np.abs(unique_times.reshape(-1, 1) - unique_times.reshape(1, -1)).reshape(-1, 1)

    def loss(log_ls):
        covs = cov_func_curry(ls=float(np.exp(log_ls[0])))(delta_t, np.zeros((1, 1))).reshape(nt, nt)
        return float(np.linalg.norm(covs - corrs))

    def loss_grad(log_ls):
        h = 1e-6
        return loss(log_ls), np.array([(loss(log_ls + h) - loss(log_ls - h)) / (2 * h)])

    res = _sp_minimize(loss_grad, np.array([0.0]), jac=True, method="L-BFGS-B")
    return float(np.exp(res.x[0]))


def per_time_nn_distances(x, times, d=None, normalize=False):
    """parameters.py:444-531 -- nearest neighbour within each time point; with `normalize` the distances of a
    time point holding n_t cells are scaled by (n_t / target)^(1/d) (parameters.py:520-528), target = the average
    count per time point (True), normalize[t] (dict) or normalize[position of t] (list / array; :436-441)."""
    x = ensure_2d(x)
    times = np.asarray(times)
    uniq = np.unique(times)
    out = np.empty(x.shape[0])
    for pos, t in enumerate(uniq):
        idx = np.flatnonzero(times == t)
        nn = exact_nn_distances(x[idx])
        if normalize is not False and normalize is not None:
            if isinstance(normalize, bool):
                target = x.shape[0] / len(uniq)
            elif isinstance(normalize, dict):
                target = normalize[t.item()]
            else:
                target = normalize[pos]
            dd = np.asarray(d, dtype=np.float64)
            nn = (len(idx) / target) ** (1.0 / (dd if dd.ndim == 0 else dd[idx])) * nn
        out[idx] = nn
    return out


# --------------------------------------------------------------------------
# synthetic workloads shared verbatim by CPU baseline, tests and bench
# --------------------------------------------------------------------------
def gaussian_mixture(n, d, seed, k=10):
    """10-component isotropic Gaussian mixture, float64, PCG64(seed)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    means = rng.normal(0.0, 3.0, size=(k, d))
    sig = rng.uniform(0.5, 1.5, size=k)
    comp = rng.integers(0, k, size=n)
    x = means[comp] + rng.normal(size=(n, d)) * sig[comp][:, None]
    return np.ascontiguousarray(x[rng.permutation(n)])
