"""Parity of every C-ABI operator against the oracle, on a real MI355X (-m gpu).

All calls go through libmellon_hip.so via mellon_amd._lib (ctypes).  Tolerances are written
next to each check: kernel values are rounding-level; factor-dependent quantities carry the
conditioning of K_uu + 1e-6 I (cond(Lp) up to ~1e4) and are checked to <= 1e-7 relative to the
largest entry, far inside the 1e-5 log-density budget of BASELINE.json.
"""
import numpy as np
import pytest
import scipy.linalg as sla

from oracle import mellon_oracle as mo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from mellon_amd import _lib
    return _lib.default_context()


def _pair(product_cov):
    """product covariance -> equivalent oracle covariance through the JSON wire format."""
    return mo.Covariance.from_dict(product_cov.to_dict())


def relmax(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_device_is_gfx950(ctx):
    info = ctx.device_info()
    assert info["arch"].startswith("gfx950") and info["n_cu"] >= 64


@pytest.mark.parametrize("name", ["Matern32", "Matern52", "ExpQuad", "Exponential", "RatQuad", "Linear"])
@pytest.mark.parametrize("shape", [(1, 1, 1), (7, 5, 3), (130, 67, 20), (300, 129, 50)])
def test_kernel_matrix_leaves(ctx, name, shape):
    from mellon_amd import cov
    n, m, d = shape
    rng = np.random.default_rng(n * 1000 + m)
    x, y = rng.normal(size=(n, d)) * 2.0, rng.normal(size=(m, d)) * 2.0
    y[: min(n, m) // 2] = x[: min(n, m) // 2]          # coincident points: the +1e-12 branch
    c = getattr(cov, name)(2.0, 1.7) if name == "RatQuad" else getattr(cov, name)(1.7)
    K = c(x, y)
    assert K.shape == (n, m)
    ref = _pair(c)(x, y)
    # coincident points evaluate sqrt(1e-12 + rounding noise of xx - 2xy + yy) (util.py:365): the
    # noise (~1e-15 |x|^2) is implementation-defined, so kernels with a cusp at 0 (Exponential)
    # differ by up to ~1e-7 there in ANY two implementations; everywhere else rounding-level agreement.
    co = np.zeros((n, m), dtype=bool)
    h = min(n, m) // 2
    co[np.arange(h), np.arange(h)] = True
    assert np.abs(K - ref)[~co].max(initial=0.0) < 1e-12 * max(np.abs(ref).max(), 1.0)
    assert np.abs(K - ref)[co].max(initial=0.0) < 1e-6


def test_kernel_matrix_algebra_and_active_dims(ctx):
    from mellon_amd import cov
    rng = np.random.default_rng(3)
    x, y = rng.normal(size=(90, 6)), rng.normal(size=(41, 6))
    a = cov.Matern52(1.3, active_dims=slice(None, -1))
    b = cov.Matern52(0.6, active_dims=-1)
    e = cov.ExpQuad(0.9, active_dims=[0, 2, 5])
    mask = cov.Matern32(1.1, active_dims=np.array([True, False, True, False, False, True]))
    for c in (a * b, a + b, a + 2.0, 3.0 * a, a ** 2, (a * b) + e * 0.5, mask,
              cov.Mul(a, cov.ExpQuad(0.9, active_dims=[0, 2, 4]), active_dims=[0, 1, 2, 3, 5])):
        assert relmax(c(x, y), _pair(c)(x, y)) < 1e-12, repr(c)


def test_custom_python_kernel_is_evaluated_block_wise(ctx):
    """A subclass with its own Python k (reference base_cov.py:17-69) does not lower to a device program: lower() hands
    back the block evaluator, k() returns the user's values, scalar algebra on top runs on the device."""
    from mellon_amd.base_cov import BlockCov, Covariance

    class Mine(Covariance):
        def k(self, x, y):
            return x @ y.T

    rng = np.random.default_rng(2)
    x, y = rng.normal(size=(40, 3)), rng.normal(size=(7, 3))
    assert isinstance((Mine() * 2.0).lower(3), BlockCov)
    assert relmax((Mine() * 2.0 + 1.0)(x, y), 2.0 * (x @ y.T) + 1.0) < 1e-14
    with pytest.raises(NotImplementedError):
        Mine().k_grad(x)(y)


@pytest.mark.parametrize("m", [1, 5, 64, 65, 200, 777])
def test_cholesky(ctx, m):
    rng = np.random.default_rng(m)
    B = rng.normal(size=(m, m + 3))
    A = B @ B.T / m + np.eye(m) * 0.1
    L = ctx.chol_lower(A, add_diag=1e-6)
    ref = sla.cholesky(A + 1e-6 * np.eye(m), lower=True)
    assert np.allclose(np.triu(L, 1), 0.0)
    assert relmax(L, ref) < 1e-11


def test_cholesky_not_pd(ctx):
    A = -np.eye(70)
    with pytest.raises(ValueError, match="not positively definite"):
        ctx.chol_lower(A, add_diag=1e-6)
    A = np.eye(70)
    A[40, 40] = np.nan
    with pytest.raises(ValueError, match="not positively definite"):
        ctx.chol_lower(A, add_diag=1e-6)


@pytest.mark.parametrize("m,p", [(5, 1), (64, 3), (129, 1), (300, 7), (700, 130)])
def test_trsm(ctx, m, p):
    rng = np.random.default_rng(m + p)
    Lf = np.tril(rng.normal(size=(m, m))) * 0.1 + np.eye(m) * 2.0
    B = rng.normal(size=(m, p))
    assert relmax(ctx.trsm_lower(Lf, B, trans=False), sla.solve_triangular(Lf, B, lower=True)) < 1e-11
    assert relmax(ctx.trsm_lower(Lf, B, trans=True), sla.solve_triangular(Lf.T, B, lower=False)) < 1e-11


def _problem(n, d, m, seed, kern="Matern52"):
    x = mo.gaussian_mixture(n, d, seed)
    nn = mo.exact_nn_distances(x)
    ls = mo.compute_ls(nn)
    mu = mo.compute_mu(nn, d)
    rng = np.random.default_rng(seed)
    xu = x[rng.choice(n, size=m, replace=False)] + 0.01 * rng.normal(size=(m, d))
    return x, nn, ls, mu, xu


@pytest.mark.parametrize("n,d,m,kern", [(1000, 10, 37, "Matern52"), (5000, 20, 256, "ExpQuad"),
                                        (20000, 50, 1000, "Matern52"), (3001, 3, 130, "Matern32")])
def test_fit_pipeline_sparse(ctx, n, d, m, kern):
    from mellon_amd import cov
    x, nn, ls, mu, xu = _problem(n, d, m, seed=n + m)
    c = getattr(cov, kern)(ls)
    oc = _pair(c)
    fit = ctx.fit_prepare(c.lower(d), x, xu, 1e-6)
    Lp_ref = mo.full_rank(xu, oc)
    L_ref = mo.standard_low_rank(x, oc, xu, Lp=Lp_ref)
    assert relmax(fit.Lp(), Lp_ref) < 1e-8          # conditioning of K_uu + 1e-6 I enters here
    L = fit.L()
    assert L.shape == (n, m)
    assert relmax(L, L_ref) < 1e-7
    # L L^T ~ K restricted to what the factor represents: check L Lp^T == K_xu instead (backward error)
    assert relmax(L @ Lp_ref.T, oc(x, xu)) < 1e-11
    # Ridge initial value (parameters.py:895-896) against the oracle on the SAME L
    target = mo.mle(nn, d) - mu
    z0 = fit.ridge_init(target)
    z0_ref = mo.compute_initial_value(nn, d, mu, L)
    assert relmax(z0, z0_ref) < 1e-7
    # objective / gradient / diagonal Hessian (inference.py:167-192, 291-338)
    V, Vdr = mo.nn_likelihood_constants(nn, d)
    fit.set_likelihood(V, Vdr, mu)
    rng = np.random.default_rng(0)
    for z in (z0_ref, z0_ref + 0.01 * rng.normal(size=m)):
        loss, grad, hess = fit.objective(z, with_hess=True)
        loss_ref, grad_ref = mo.loss_and_grad(z, L, mu, V, Vdr)
        assert abs(loss - loss_ref) / abs(loss_ref) < 1e-12
        assert relmax(grad, grad_ref) < 1e-10
        a = np.exp(L @ z + mu + V)
        assert relmax(hess, 1.0 + (L * L).T @ a) < 1e-10
        loss2, grad2 = fit.objective(z)
        assert loss2 == loss and np.array_equal(grad2, grad)      # deterministic reductions
    # transform + predictor weights + fused predict (conditional.py:818, 899-906)
    z = z0_ref
    f = fit.transform(z, mu)
    assert relmax(f, L @ z + mu) < 1e-12
    w = fit.weights_cholesky(z)
    assert relmax(w, sla.solve_triangular(Lp_ref.T, z, lower=False)) < 1e-7
    pred = ctx.predict_mean(c.lower(d), x[: n // 3], xu, w, mu)
    assert relmax(pred, f[: n // 3]) < 1e-8           # predict(X) == fit_predict(X)  (test_density_estimator.py:40-44)
    st = fit.stage_times()
    assert st["objective_launches"] == 4 and st["objective_bytes_per_launch"] >= n * m * 8


def test_fit_pipeline_full(ctx):
    from mellon_amd import cov
    n, d = 700, 5
    x = mo.gaussian_mixture(n, d, 11)
    nn = mo.exact_nn_distances(x)
    ls, mu = mo.compute_ls(nn), mo.compute_mu(nn, d)
    c = cov.Matern52(ls)
    oc = _pair(c)
    fit = ctx.fit_prepare(c.lower(d), x, None, 1e-6)
    Lp_ref = mo.full_rank(x, oc)
    assert relmax(fit.Lp(), Lp_ref) < 1e-8
    assert relmax(fit.L(), Lp_ref) < 1e-8
    V, Vdr = mo.nn_likelihood_constants(nn, d)
    fit.set_likelihood(V, Vdr, mu)
    z = mo.compute_initial_value(nn, d, mu, Lp_ref)
    assert relmax(fit.ridge_init(mo.mle(nn, d) - mu), z) < 1e-7
    loss, grad = fit.objective(z)
    loss_ref, grad_ref = mo.loss_and_grad(z, Lp_ref, mu, V, Vdr)
    assert abs(loss - loss_ref) / abs(loss_ref) < 1e-11 and relmax(grad, grad_ref) < 1e-9
    y = Lp_ref @ z + mu
    w = fit.weights_full(y, mu)
    w_ref = sla.solve_triangular(Lp_ref.T, sla.solve_triangular(Lp_ref, y - mu, lower=True), lower=False)
    assert relmax(w, w_ref) < 1e-6
    pred = ctx.predict_mean(c.lower(d), x, x, w, mu)
    assert relmax(pred, y) < 1e-6


def test_given_Lp_is_used(ctx):
    from mellon_amd import cov
    x, nn, ls, mu, xu = _problem(800, 4, 50, seed=5)
    c = cov.Matern52(ls)
    Lp = mo.full_rank(xu, _pair(c)) * 1.0
    fit = ctx.fit_prepare(c.lower(4), x, xu, 1e-6, Lp=Lp)
    assert np.array_equal(fit.Lp(), Lp)
    assert relmax(fit.L(), mo.standard_low_rank(x, _pair(c), xu, Lp=Lp)) < 1e-7


def test_not_pd_maps_to_value_error(ctx):
    from mellon_amd import cov
    x = np.zeros((40, 2))
    with pytest.raises(ValueError, match="not positively definite"):
        ctx.fit_prepare((cov.Matern52(1.0) * -1.0).lower(2), x, x[:10], 1e-6)


@pytest.mark.parametrize("p", [1, 3, 200])
def test_predict_mean_batched(ctx, p):
    from mellon_amd import cov
    rng = np.random.default_rng(p)
    x, xu = rng.normal(size=(1500, 7)), rng.normal(size=(333, 7))
    W = rng.normal(size=(333, p)) if p > 1 else rng.normal(size=333)
    c = cov.Matern52(2.0)
    out = ctx.predict_mean(c.lower(7), x, xu, W, -1.5)
    ref = -1.5 + _pair(c)(x, xu) @ W
    assert out.shape == ref.shape and relmax(out, ref) < 1e-12


@pytest.mark.parametrize("n,m,p", [(600, 40, 3), (9000, 300, 17)])
def test_sparse_solve_function_estimator(ctx, n, m, p):
    from mellon_amd import cov
    rng = np.random.default_rng(n)
    x = mo.gaussian_mixture(n, 6, seed=n)
    xu = x[rng.choice(n, m, replace=False)]
    y = np.sin(x @ rng.normal(size=(6, p))) + 0.1 * rng.normal(size=(n, p))
    ls = mo.compute_ls(mo.exact_nn_distances(x))
    c = cov.Matern52(ls)
    W = ctx.sparse_solve(c.lower(6), x, xu, y, 0.25, 0.3, 1e-6)
    ref = mo.landmarks_conditional(x, xu, y, 0.25, _pair(c), sigma=0.3)
    # weights themselves are ill-conditioned (Lp^-T); compare the predictions they produce
    out = ctx.predict_mean(c.lower(6), x[:500], xu, W, 0.25)
    assert relmax(out, ref(x[:500])) < 1e-7


def test_reference_golden_function_estimator_on_gpu(ctx):
    """The reference's own golden vectors (tests/test_reference_results.py:26-63,93-130) through
    the HIP path: full GP (weights_full) and sparse (sparse_solve), atol 1e-5 as in the reference."""
    import json
    import os
    from mellon_amd import cov
    from oracle import jax_prng as jp
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_results.json")))
    k1, k2, k3 = jp.split(jp.prng_key(42), 3, True)
    X, y, Xt = jp.normal64(k1, (50, 2), True), jp.normal64(k2, (50, 3), True), jp.normal64(k3, (10, 2), True)
    ls = mo.compute_ls(mo.exact_nn_distances(X))
    c = cov.Matern52(ls)
    # full: L = chol(K + sigma^2 I), sigma = 1  (conditional.py:253-264 with y_is_mean=False)
    fit = ctx.fit_prepare(c.lower(2), X, None, 1.0)
    w = fit.weights_full(y, 0.0)
    pred = ctx.predict_mean(c.lower(2), Xt, X, w, 0.0)
    assert np.allclose(pred, np.array(gold["full"]["expected_pred"]), atol=1e-5)
    # sparse, 15 k-means landmarks (shared input: sklearn k_means, random_state=42)
    xu = mo.compute_landmarks(X, mo.SPARSE_CHOLESKY, 15, 42)
    W = ctx.sparse_solve(c.lower(2), X, xu, y, 0.0, 1.0, 1e-6)
    pred = ctx.predict_mean(c.lower(2), Xt, xu, W, 0.0)
    assert np.allclose(pred, np.array(gold["sparse"]["expected_pred"]), atol=1e-5)


def test_preconditioned_objective(ctx):
    """z = C^-T u with C C^T = L^T L + I: loss identical, gradient = C^-1 grad_z, round trips."""
    from mellon_amd import cov
    n, d, m = 6000, 12, 300
    x, nn, ls, mu, xu = _problem(n, d, m, seed=77)
    c = cov.Matern52(ls)
    fit = ctx.fit_prepare(c.lower(d), x, xu, 1e-6)
    V, Vdr = mo.nn_likelihood_constants(nn, d)
    fit.set_likelihood(V, Vdr, mu)
    L = fit.L()
    C = sla.cholesky(L.T @ L + np.eye(m), lower=True)
    z = fit.ridge_init(mo.mle(nn, d) - mu) + 0.01 * np.random.default_rng(1).normal(size=m)
    u = fit.precond_apply(0, z)
    assert relmax(u, C.T @ z) < 1e-9
    assert relmax(fit.precond_apply(1, u), z) < 1e-9
    loss_u, grad_u, z_back = fit.objective_precond(u)
    loss_z, grad_z = fit.objective(z_back)
    assert relmax(z_back, z) < 1e-9
    assert abs(loss_u - loss_z) <= 1e-12 * abs(loss_z)
    assert relmax(grad_u, sla.solve_triangular(C, grad_z, lower=True)) < 1e-8
    assert relmax(fit.precond_apply(2, grad_z), sla.solve_triangular(C, grad_z, lower=True)) < 1e-8


def test_implicit_mode_matches_explicit(ctx):
    """MLN_FIT_IMPLICIT streams K and folds Lp^-T into the m-vectors: every quantity must agree
    with the explicit-L handle (same maths, different association)."""
    from mellon_amd import cov
    n, d, m = 9000, 15, 400
    x, nn, ls, mu, xu = _problem(n, d, m, seed=91)
    c = cov.Matern52(ls)
    V, Vdr = mo.nn_likelihood_constants(nn, d)
    target = mo.mle(nn, d) - mu
    fe = ctx.fit_prepare(c.lower(d), x, xu, 1e-6)
    fi = ctx.fit_prepare(c.lower(d), x, xu, 1e-6, implicit=True)
    assert fi.stage_times()["trsm_s"] == 0.0
    assert relmax(fi.L(100, 700), fe.L(100, 700)) < 1e-9              # rows materialised on demand
    for f in (fe, fi):
        f.set_likelihood(V, Vdr, mu)
    z0e, z0i = fe.ridge_init(target), fi.ridge_init(target)
    # the implicit Gram is Lp^-1 (K^T K) Lp^-T (rounding amplified by |Lp^-1|^2): an initial guess /
    # preconditioner, not a result -- agreement to 1e-3 is plenty (the optimum does not depend on it)
    assert relmax(z0i, z0e) < 1e-3
    z = z0e + 0.01 * np.random.default_rng(0).normal(size=m)
    (le, ge), (li, gi) = fe.objective(z), fi.objective(z)
    assert abs(li - le) < 1e-10 * abs(le) and relmax(gi, ge) < 1e-7
    assert relmax(fi.transform(z, mu), fe.transform(z, mu)) < 1e-9
    u = fe.precond_apply(0, z)
    lue, gue, ze = fe.objective_precond(u)
    lui, gui, zi = fi.objective_precond(fi.precond_apply(0, z))
    assert abs(lui - lue) < 1e-9 * abs(lue) and relmax(zi, ze) < 1e-7
    # each handle has its own C: compare the gradients back in z coordinates, g_z = C^T g_u
    assert relmax(fi.precond_apply(0, gui), fe.precond_apply(0, gue)) < 1e-6
    with pytest.raises(NotImplementedError):
        fi.objective(z, with_hess=True)


def test_subsampled_gram_preconditioner(ctx):
    """A Gram from every k-th cell is a valid preconditioner: same optimum, similar pass count."""
    from mellon_amd import cov
    n, d, m = 40000, 10, 200
    x, nn, ls, mu, xu = _problem(n, d, m, seed=17)
    c = cov.Matern52(ls)
    V, Vdr = mo.nn_likelihood_constants(nn, d)
    opts = dict(maxiter=5000, maxcor=30, ftol=1e-13, gtol=1e-7)
    out = []
    for stride in (1, 20):
        f = ctx.fit_prepare(c.lower(d), x, xu, 1e-6, implicit=True)
        f.set_likelihood(V, Vdr, mu)
        f.precond_build(stride)
        z0 = f.ridge_init(mo.mle(nn, d) - mu)
        calls = []
        def fun(u, f=f, calls=calls):
            l, g, z = f.objective_precond(u)
            calls.append(z)
            return l, g
        res = mo.minimize_lbfgsb(fun, f.precond_apply(0, z0), opts)
        out.append((f.transform(f.precond_apply(1, res.pre_transformation), mu), res.n_eval))
    (fa, ea), (fb, eb) = out
    assert relmax(fb, fa) < 1e-6
    assert eb <= ea + 10


def test_rccl_single_rank_communicator():
    """The RCCL binding (dlopen, ncclGetUniqueId / CommInitRank / AllReduce, fp64 sum) on one GPU:
    a 1-rank communicator must leave every result unchanged.  Uses its own context."""
    from mellon_amd import _lib, cov
    c2 = _lib.Context(0)
    try:
        uid = c2.comm_unique_id()
        assert len(uid) == 128
        c2.comm_init(uid, 1, 0)
        v = np.arange(7.0)
        assert np.array_equal(c2.allreduce_sum(v), v)
        n, d, m = 3000, 6, 64
        x, nn, ls, mu, xu = _problem(n, d, m, seed=8)
        kern = cov.Matern52(ls)
        V, Vdr = mo.nn_likelihood_constants(nn, d)
        a = c2.fit_prepare(kern.lower(d), x, xu, 1e-6, implicit=True)
        b = _lib.default_context().fit_prepare(kern.lower(d), x, xu, 1e-6, implicit=True)
        for f in (a, b):
            f.set_likelihood(V, Vdr, mu)
        za, zb = a.ridge_init(mo.mle(nn, d) - mu), b.ridge_init(mo.mle(nn, d) - mu)
        assert np.array_equal(za, zb)
        (la, ga), (lb, gb) = a.objective(za), b.objective(zb)
        assert la == lb and np.array_equal(ga, gb)
        a.close()
    finally:
        c2.close()


def test_kmeans_landmarks(ctx):
    """mln_kmeans: k-means++ / Lloyd on the device -- same quality as sklearn's k_means (inertia within
    a few %), every centre is the mean of its members, reproducible to rounding for a fixed seed."""
    from sklearn.cluster import k_means
    x = mo.gaussian_mixture(20000, 12, seed=33)
    m = 200
    c, n_iter, inertia = ctx.kmeans(x, m, seed=42, return_info=True)
    assert c.shape == (m, 12) and 1 <= n_iter <= 300
    _, _, ref_inertia = k_means(x, m, n_init=1, random_state=42)
    assert inertia < 1.05 * ref_inertia
    d2 = ((x[:, None, :] - c[None, :, :]) ** 2).sum(-1) if x.shape[0] * m < 5e6 else None
    lab = np.argmin(mo.distance(x, c), axis=1)
    assert abs(np.sum(mo.distance(x, c)[np.arange(len(x)), lab] ** 2) - inertia) < 1e-6 * inertia
    c2 = ctx.kmeans(x, m, seed=42)
    assert relmax(c2, c) < 1e-9                      # same seed -> same centres up to atomic summation order
    assert relmax(ctx.kmeans(x, m, seed=7), c) > 1e-3  # different seed -> different seeding
    # degenerate: as many clusters as points
    small = x[:50]
    cs = ctx.kmeans(small, 50, seed=1)
    assert np.allclose(np.sort(cs, axis=0), np.sort(small, axis=0))


@pytest.mark.parametrize("m", [2100, 6000, 8100])
def test_objective_wide_landmark_counts(ctx, m):
    """Every column-per-thread specialisation of k_objective (m up to the 8192 limit) against NumPy."""
    from mellon_amd import _lib
    rng = np.random.default_rng(m)
    n = 1500
    L = rng.normal(size=(n, m)) * (0.5 / np.sqrt(m))
    nn = rng.uniform(0.2, 1.0, size=n)
    V, Vdr = mo.nn_likelihood_constants(nn, 5)
    fit = _lib.Fit.from_L(ctx, L)
    fit.set_likelihood(V, Vdr, -3.0)
    z = rng.normal(size=m) * 0.3
    loss, grad, hess = fit.objective(z, with_hess=True)
    loss_ref, grad_ref = mo.loss_and_grad(z, L, -3.0, V, Vdr)
    a = np.exp(L @ z - 3.0 + V)
    assert abs(loss - loss_ref) < 1e-12 * abs(loss_ref)
    assert relmax(grad, grad_ref) < 1e-11 and relmax(hess, 1.0 + (L * L).T @ a) < 1e-11
    assert relmax(fit.transform(z, -3.0), L @ z - 3.0) < 1e-12
    u = fit.precond_apply(0, z)
    lu, gu, zb = fit.objective_precond(u)
    assert abs(lu - loss_ref) < 1e-10 * abs(loss_ref) and relmax(zb, z) < 1e-9


# ---- symmetric eigensolver (mln_eigh) and the Nystroem factors (decomposition.py:23-76,126-171,213-266) ----
def _kernel_like(m, seed, jitter=1e-6):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(m, 4))
    xx = (x * x).sum(1)
    d2 = np.maximum(xx[:, None] - 2 * x @ x.T + xx[None, :], 0.0)
    return np.exp(-0.5 * d2 / 4.0) + jitter * np.eye(m)     # spectrum spans >= 6 decades, heavy clustering


@pytest.mark.parametrize("m", [1, 2, 5, 16, 17, 33, 100, 257, 600])
def test_eigh_kernel_matrices(ctx, m):
    A = _kernel_like(m, m)
    w, V = ctx.eigh(A)
    w_ref = np.linalg.eigvalsh(A)
    scale = np.abs(w_ref).max()
    assert np.all(np.diff(w) >= 0)                                        # ascending, LAPACK order
    assert np.abs(w - w_ref).max() < 1e-11 * scale                        # absolute accuracy ~ eps |A|
    assert np.abs(V.T @ V - np.eye(m)).max() < 1e-11                      # product of plane rotations
    assert np.abs(A @ V - V * w).max() < 1e-11 * scale                    # residual


def test_eigh_special_matrices(ctx):
    rng = np.random.default_rng(3)
    # indefinite, with a repeated eigenvalue and a zero eigenvalue
    Q, _ = np.linalg.qr(rng.normal(size=(40, 40)))
    lam = np.concatenate([[-3.0, -3.0, 0.0], rng.uniform(-1, 5, 37)])
    A = (Q * lam) @ Q.T
    w, V = ctx.eigh(A)
    assert np.abs(w - np.sort(lam)).max() < 1e-12 and np.abs(A @ V - V * w).max() < 1e-12
    # identity (all eigenvalues equal), zero matrix, diagonal matrix: no rotation at all
    for D in (np.eye(20), np.zeros((20, 20)), np.diag(np.arange(20.0)[::-1])):
        w, V = ctx.eigh(D)
        assert np.array_equal(w, np.sort(np.diag(D))) and np.abs(V.T @ V - np.eye(20)).max() < 1e-15
    # only the symmetric part counts (jax eigh symmetrize_input default)
    B = rng.normal(size=(30, 30))
    w, _ = ctx.eigh(B)
    assert np.abs(w - np.linalg.eigvalsh(0.5 * (B + B.T))).max() < 1e-12
    with pytest.raises(Exception):
        ctx.eigh(np.full((4, 4), np.nan))


@pytest.mark.parametrize("rank", [0.99, 0.9999, 1.0, 7, 40, 1000])
def test_modified_low_rank_matches_reference_algorithm(ctx, rank):
    from mellon_amd import cov
    from mellon_amd.decomposition import _modified_low_rank
    x = mo.gaussian_mixture(700, 6, 21)
    xu = x[np.random.default_rng(2).choice(700, 64, replace=False)]
    k = cov.Matern52(ls=2.5)
    L = np.asarray(_modified_low_rank(x, k, xu, rank=rank, jitter=1e-6))
    ref = mo.modified_low_rank(x, _pair(k), xu, rank=rank, jitter=1e-6)      # QR + two eigh, as the reference
    assert L.shape == ref.shape
    G, Gr = L @ L.T, ref @ ref.T
    assert np.abs(G - Gr).max() < 1e-10 * np.abs(Gr).max()
    # columns are eigen-directions in ascending order: equal up to sign wherever the eigenvalue is isolated
    assert np.abs((L * L).sum(0) - (ref * ref).sum(0)).max() < 1e-9 * (ref * ref).sum(0).max()


@pytest.mark.parametrize("rank", [0.99, 12, 1.0])
def test_full_decomposition_low_rank_matches_reference_algorithm(ctx, rank):
    from mellon_amd import cov
    from mellon_amd.decomposition import _full_decomposition_low_rank
    x = mo.gaussian_mixture(300, 5, 22)
    k = cov.ExpQuad(ls=2.0)
    L = np.asarray(_full_decomposition_low_rank(x, k, rank=rank, jitter=1e-6))
    ref = mo.full_decomposition_low_rank(x, _pair(k), rank=rank, jitter=1e-6)
    assert L.shape == ref.shape
    assert np.abs(L @ L.T - ref @ ref.T).max() < 1e-10 * np.abs(ref @ ref.T).max()


def test_gram_eigh_explicit_and_implicit_factor(ctx):
    from mellon_amd import cov
    x = mo.gaussian_mixture(200, 3, 1)
    fit = ctx.fit_prepare(cov.Matern52(ls=1.5).lower(3), x, x[:20], 1e-6, implicit=True)
    Si = fit.gram_eigh()                                      # eigenvalues of Lp^-1 (K^T K) Lp^-T: the same spectrum
    with pytest.raises(NotImplementedError):
        fit.project(5)                                        # projecting needs the explicit rows
    fit2 = ctx.fit_prepare(cov.Matern52(ls=1.5).lower(3), x, x[:20], 1e-6)
    with pytest.raises(Exception):
        fit2.project(5)                                       # no eigenvectors yet
    S = fit2.gram_eigh()
    Lb = fit2.L()
    assert np.abs(S - np.linalg.eigvalsh(Lb.T @ Lb)).max() < 1e-11 * S.max()
    assert np.abs(Si - S).max() < 1e-6 * S.max()
    with pytest.raises(ValueError):
        fit2.project(21)
    p5 = fit2.project(5)
    assert p5.m == 5 and p5.L().shape == (200, 5)


# ---- analytic gradients: Covariance.k_grad and Predictor.gradient -----------------------------------------
_ACTIVE_DIMS = [None, slice(2), 1, slice(None, None, 2), [1, 2]]


@pytest.mark.parametrize("name", ["Matern32", "Matern52", "ExpQuad", "Exponential", "RatQuad", "Linear"])
@pytest.mark.parametrize("active_dims", _ACTIVE_DIMS)
def test_k_grad_leaves(ctx, name, active_dims):
    # tests/test_cov.py:21-64 (x = 1, y in {2, 1.5}, ls = 1.2)
    from mellon_amd import cov
    cls = getattr(cov, name)
    k = cls(3, 1.2, active_dims=active_dims) if name == "RatQuad" else cls(1.2, active_dims=active_dims)
    x = np.ones((5, 4))
    y = np.ones((6, 4)) * 2
    y[1] = 1.5
    g = k.k_grad(x)(y)
    ref = _pair(k).k_grad(x)(y)
    assert g.shape == (5, 6, 4)
    assert np.abs(g - ref).max() < 1e-13 * max(np.abs(ref).max(), 1.0)
    # random, non-degenerate points and coincident points (distance floor 1e-6, delta = 0)
    rng = np.random.default_rng(1)
    x2, y2 = rng.normal(size=(37, 4)), rng.normal(size=(29, 4))
    y2[:5] = x2[:5]
    g2, r2 = k.k_grad(x2)(y2), _pair(k).k_grad(x2)(y2)
    assert np.abs(g2 - r2).max() < 1e-9 * max(np.abs(r2).max(), 1.0)


@pytest.mark.parametrize("active_dims", _ACTIVE_DIMS)
def test_k_grad_composites(ctx, active_dims):
    # tests/test_base_cov.py:41-53,87-99,132-144,173-185
    from mellon_amd import cov
    x = np.ones((2, 3))
    for k in (cov.Add(cov.Matern32(1.4), cov.Exponential(3.4), active_dims=active_dims),
              cov.Mul(cov.Matern32(1.4), cov.Exponential(3.4), active_dims=active_dims),
              cov.Pow(cov.Matern32(1.4), 3.2, active_dims=active_dims)):
        g, ref = k.k_grad(x)(2 * x), _pair(k).k_grad(x)(2 * x)
        assert np.abs(g - ref).max() < 1e-13
    h = 0.2 + 1.1 * cov.Matern52(1.4, active_dims=0) + \
        2.1 * cov.Exponential(3.4, active_dims=[1, 2]) * cov.RatQuad(1.1, 3.4, active_dims=slice(0, 2, 1)) + \
        cov.Matern52(1.0, active_dims=[False, True, True])
    rng = np.random.default_rng(2)
    xr, yr = rng.normal(size=(40, 3)), rng.normal(size=(33, 3))
    g, ref = h.k_grad(xr)(yr), _pair(h).k_grad(xr)(yr)
    assert np.abs(g - ref).max() < 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("n,m,d", [(1, 1, 1), (130, 70, 3), (300, 257, 20), (129, 64, 50)])
def test_predict_gradient_matches_contracted_k_grad(ctx, n, m, d):
    from mellon_amd import cov
    rng = np.random.default_rng(n + m + d)
    xq, c, w = rng.normal(size=(n, d)), rng.normal(size=(m, d)), rng.normal(size=m)
    kernels = [cov.Matern52(1.7 * np.sqrt(d)), cov.ExpQuad(2.0 * np.sqrt(d))]
    if d >= 3:
        kernels.append(cov.Matern52(1.5 * np.sqrt(d), active_dims=slice(None, -1)) * cov.ExpQuad(1.2, active_dims=-1))
        kernels.append(cov.Matern32(2.0, active_dims=[0, 2]) + 0.5 * cov.Linear(3.0))
        # composite programs of stationary leaves take the one-coefficient-matrix-per-leaf GEMM route above 2^16 pairs
        kernels.append((cov.Matern52(2.0 * np.sqrt(d)) * cov.RatQuad(1.5, 3.0 * np.sqrt(d))) ** 2.0)
        kernels.append(cov.Matern52(2.0, active_dims=slice(0, 2)) * cov.ExpQuad(1.0, active_dims=slice(1, 3))
                       + 0.3 * cov.Matern32(0.7 * np.sqrt(d)))
    for k in kernels:
        g = ctx.predict_gradient(k.lower(d), xq, c, w)
        ref = np.einsum("j,jik->ik", w, _pair(k).k_grad(c)(xq))     # d k(x_i, c_j)/dx_i = k_grad(c)(x)[j, i]
        assert g.shape == (n, d)
        assert np.abs(g - ref).max() < 1e-10 * max(np.abs(ref).max(), 1e-300)


@pytest.mark.parametrize("n,m,d", [(1, 1, 1), (70, 40, 3), (130, 257, 6), (40, 64, 20)])
def test_predict_hessian_matches_oracle(ctx, n, m, d):
    """mln_predict_hessian (closed-form second derivatives, one GEMM per leaf pair) against the oracle's
    finite-difference derivative of the analytic k_grad contraction -- every leaf kind, sums, products, powers,
    overlapping and disjoint active dims, a Linear factor."""
    from mellon_amd import cov
    rng = np.random.default_rng(n + m + d)
    xq, c, w = rng.normal(size=(n, d)), rng.normal(size=(m, d)), rng.normal(size=m)
    s = np.sqrt(d)
    kernels = [cov.Matern52(1.7 * s), cov.Matern32(1.9 * s), cov.ExpQuad(2.0 * s), cov.Exponential(2.5 * s),
               cov.RatQuad(1.5, 2.2 * s)]
    if d >= 3:
        kernels += [cov.Matern52(1.5 * s, active_dims=slice(None, -1)) * cov.Matern52(1.2, active_dims=-1),
                    cov.Matern32(2.0, active_dims=[0, 2]) + 0.5 * cov.ExpQuad(3.0 * s),
                    (cov.Matern52(2.0 * s) * cov.RatQuad(1.5, 3.0 * s)) ** 2.0,
                    cov.Matern52(2.0, active_dims=slice(0, 2)) * cov.ExpQuad(1.0, active_dims=slice(1, 3))
                    + 0.3 * cov.Matern32(0.7 * s),
                    cov.Linear(2.0) * cov.Matern52(1.5 * s) + cov.Linear(3.0, active_dims=[0, 1])]
    for k in kernels:
        H = ctx.predict_hessian(k.lower(d), xq, c, w)
        ref = mo.Predictor(_pair(k), c, w, 0.0, m).hessian(xq)
        assert H.shape == (n, d, d)
        assert np.abs(H - np.swapaxes(H, 1, 2)).max() <= 1e-12 * max(np.abs(H).max(), 1e-300)
        # kernels of low smoothness at the centres (Exponential: a cusp; Matern32: discontinuous third derivative) limit
        # the finite-difference oracle, not the closed forms
        tol = 2e-5 if "Exponential" in repr(k) else 2e-6 if "Matern32" in repr(k) else 2e-7
        assert np.abs(H - ref).max() < tol * max(np.abs(ref).max(), 1e-300), (repr(k), np.abs(H - ref).max(), np.abs(ref).max())


def test_util_helpers_distance_and_rank(ctx):
    """util.distance (util.py:351-366) via the value-only DISTANCE leaf; util.test_rank = matrix_rank(L, rtol) from the
    device's Gram eigenvalues (util.py:429-483); stabilize / add_variance (util.py:269-331)."""
    from mellon_amd import util
    rng = np.random.default_rng(0)
    x, y = rng.normal(size=(77, 5)), rng.normal(size=(33, 5))
    y[:4] = x[:4]
    # coincident rows: sqrt(1e-12 + cancellation residue of |x|^2 - 2 x.y + |y|^2), equal to ~1e-9 only
    np.testing.assert_allclose(util.distance(x, y), mo.distance(x, y), rtol=1e-12, atol=5e-9)
    big = rng.normal(size=(5000, 7))
    np.testing.assert_allclose(util.distance(big, big[:300]), mo.distance(big, big[:300]), rtol=1e-11, atol=1e-7)
    L = rng.normal(size=(400, 12)) @ np.diag([10, 9, 8, 6, 5.5, 4, 1, 0.5, 0.2, 0.1, 0.01, 1e-4])
    for tol in (0.5, 0.05, 1e-3):
        assert util.test_rank(L, tol=tol, threshold=0.8) == np.linalg.matrix_rank(L, rtol=tol)
    assert util.test_rank(L.T, tol=0.5, threshold=0.8) == np.linalg.matrix_rank(L, rtol=0.5)
    K = rng.normal(size=(6, 6)); K = K @ K.T
    np.testing.assert_allclose(util.stabilize(K, 1e-3), mo.stabilize(K, 1e-3))
    M = rng.normal(size=(6, 2)) * 1e-4
    np.testing.assert_allclose(util.add_variance(K, M), mo.add_variance(K, M))
    np.testing.assert_allclose(util.add_variance(K, 0.3), mo.add_variance(K, 0.3))
    with pytest.raises(Exception):
        ctx.predict_gradient(__import__("mellon_amd").base_cov.LoweredCov([(7, 1.0, 1.0, np.arange(5))], [(0, 0, 0.0)]),
                             x, y, np.ones(33))


@pytest.mark.parametrize("name", ["Matern32", "Matern52", "ExpQuad", "Exponential", "RatQuad"])
@pytest.mark.parametrize("d", [8, 50, 60])
def test_kernel_matrix_persistent_rows(ctx, name, d):
    """The persistent-row kernel of the fit (cov_rows_impl.h: n >= 4096, m >= 256, one stationary leaf over all
    d <= 64 columns) with its straight-line sqrt / exp: ragged last row block and last centre tile, coincident
    points, and distances from 0 to far beyond the length scale (exp underflow), all three k-step variants."""
    from mellon_amd import cov
    n, m = 4500, 300
    rng = np.random.default_rng(d)
    x = rng.normal(size=(n, d)) * 1.5
    y = rng.normal(size=(m, d)) * 1.5
    y[:100] = x[:100]                                   # coincident points: the +1e-12 branch
    x[200:260] *= 400.0                                 # r >> 1: e^-r underflows
    y[250:] = x[300:350] + 1e-3 * rng.normal(size=(50, d))   # r << 1
    ls = 1.7 * np.sqrt(d / 8.0)                         # typical pair distances of a few length scales at every d
    c = getattr(cov, name)(2.0, ls) if name == "RatQuad" else getattr(cov, name)(ls)
    K = c(x, y)
    ref = _pair(c)(x, y)
    # (near-)coincident pairs: the cancellation in xx - 2xy + yy, not the sqrt / exp code, sets the error there
    # (see test_kernel_matrix_leaves)
    co = mo.distance(x, y) < 0.05
    assert co.sum() >= 150
    assert np.all(np.isfinite(K))
    assert np.abs(K - ref)[~co].max() < 1e-12
    assert np.abs(K - ref)[co].max() < 1e-6
    big = (ref > 1e-3) & ~co
    assert big.sum() > 10000
    assert (np.abs(K - ref)[big] / ref[big]).max() < 1e-11       # (summation order of the d-term dot products differs)


def _digit_gram_exact(a):
    """What csrc/gram_i8.hip computes, in exact int64 arithmetic: three balanced base-256 digits of
    q = round(a * 8355711) and all nine digit-pair products -- i.e. q^T q."""
    scale = 8355711
    q = np.rint(np.clip(a, 0.0, 1.0) * scale).astype(np.int64)
    d0 = ((q + 128) & 255) - 128
    q1 = (q - d0) >> 8
    d1 = ((q1 + 128) & 255) - 128
    d2 = (q1 - d1) >> 8
    assert d2.min() >= 0 and d2.max() <= 127 and np.array_equal(d0 + 256 * d1 + 65536 * d2, q)
    D = [d.astype(np.float64) for d in (d0, d1, d2)]          # |sum| <= 2^14 rows < 2^53: the BLAS product is exact
    acc = np.zeros((a.shape[1], a.shape[1]), dtype=np.int64)
    for i in range(3):
        for j in range(3):
            acc += (D[i].T @ D[j]).astype(np.int64) << (8 * (i + j))
    return acc, q, scale


@pytest.mark.parametrize("rows,m", [(1000, 300), (4096, 512), (70_001, 257)])
def test_integer_gram_of_the_preconditioner(ctx, rows, m):
    """The int8-MFMA Gram (three signed digit planes, int32 accumulation per k-chunk): integer-exact against int64
    NumPy -- the only floating-point operations are the final weighted sum of three exact integers and the sum of the
    k-chunks -- including ragged shapes, more than one k-chunk, and the values 0 and 1 themselves."""
    rng = np.random.default_rng(rows + m)
    a = rng.random((rows, m)) ** 3
    a[rng.integers(0, rows, 50), rng.integers(0, m, 50)] = 1.0
    a[rng.integers(0, rows, 50), rng.integers(0, m, 50)] = 0.0
    got, _ = ctx.diag_gram_i8(a)
    acc, q, scale = _digit_gram_exact(a)
    want = acc.astype(np.float64) / float(scale) ** 2
    assert np.array_equal(got, got.T)
    assert np.abs(got - want).max() <= 4e-16 * np.abs(want).max() * max(1, rows // 32768 + 1)
    assert np.array_equal(acc, (q.T.astype(np.float64) @ q.astype(np.float64)).astype(np.int64)) or rows > 500   # 2^46 rows < 2^53
    assert np.linalg.eigvalsh(got).min() >= -1e-12 * np.abs(got).max()                 # a Gram: positive semi-definite
    exact = a.T @ a
    assert np.abs(got - exact).max() <= 2.0 ** -20 * np.abs(exact).max()


@pytest.mark.parametrize("m", [1, 2, 3, 65, 700])
def test_rank_count_by_sturm_sequences(ctx, m):
    """mln_fit_gram_rank (Householder tridiagonalisation of L^T L + Sturm counts, csrc/tridiag.hip) == the count from the
    eigenvalues (mln_fit_gram_eigh) == numpy.linalg.matrix_rank(L, rtol) -- for a spectrum with clusters, repeated and
    tiny singular values, thresholds on both sides of every gap."""
    from mellon_amd import _lib
    rng = np.random.default_rng(100 + m)
    n = max(3 * m, 8)
    q1, _ = np.linalg.qr(rng.normal(size=(n, m)))
    q2, _ = np.linalg.qr(rng.normal(size=(m, m)))
    s = np.sort(np.concatenate([[1.0], 10.0 ** rng.uniform(-6, 0, size=max(m - 1, 0))]))[::-1][:m]
    if m >= 65:
        s[5:9] = s[5]                    # a repeated singular value
        s[-3:] = 1e-12                   # numerically rank deficient
    L = (q1 * s) @ q2.T
    fit = _lib.Fit.from_L(ctx, np.ascontiguousarray(L))
    try:
        ev = np.sqrt(np.maximum(fit.gram_eigh(), 0.0))
        for tol in (0.5, 0.3, 0.05, 1e-3, 1e-5):
            # thresholds at least 1 % away from a singular value (the Gram squares the conditioning: ties are undefined)
            if np.any(np.abs(s / s.max() - tol) < 0.01 * tol):
                continue
            rank, smax = fit.gram_rank(tol)
            assert rank == int(np.count_nonzero(ev > tol * ev.max())) == np.linalg.matrix_rank(L, rtol=tol), (m, tol)
            assert abs(smax - s.max()) < 1e-9 * s.max()
            assert fit.stage_times()["rank_path"] == 1          # counted by inertia (csrc/ldl_inertia.hip)
    finally:
        fit.close()


@pytest.mark.parametrize("m", [130, 700])
def test_rank_count_paths_agree(ctx, m, monkeypatch):
    """The count from the inertia of G - x I (signed block factorisation, no pivoting, guarded by its smallest pivot) and the
    count from the tridiagonalised Gram are the same number, also for a Gram with negative entries, a threshold inside a
    cluster of eigenvalues' gap, and a rank-one Gram (the shape the kernel matrices of a fit have)."""
    from mellon_amd import _lib
    rng = np.random.default_rng(7 + m)
    n = 3 * m
    cases = [rng.normal(size=(n, m)) @ np.diag(10.0 ** rng.uniform(-4, 0, size=m)),
             np.outer(rng.uniform(1, 2, size=n), rng.uniform(1, 2, size=m)) + 1e-3 * rng.normal(size=(n, m)),
             np.abs(rng.normal(size=(n, m))) + 0.5]
    for L in cases:
        fit = _lib.Fit.from_L(ctx, np.ascontiguousarray(L))
        try:
            for tol in (0.5, 0.1, 1e-2):
                sv = np.linalg.svd(L, compute_uv=False)
                if np.any(np.abs(sv / sv.max() - tol) < 0.01 * tol):
                    continue
                r1, s1 = fit.gram_rank(tol)
                assert fit.stage_times()["rank_path"] == 1
                monkeypatch.setenv("MELLON_AMD_RANK_LDL", "0")
                r2, s2 = fit.gram_rank(tol)
                monkeypatch.delenv("MELLON_AMD_RANK_LDL")
                assert fit.stage_times()["rank_path"] == 2
                assert r1 == r2 == np.linalg.matrix_rank(L, rtol=tol), (m, tol)
                assert abs(s1 - sv.max()) < 1e-9 * sv.max() and abs(s2 - s1) < 1e-9 * s1
        finally:
            fit.close()
