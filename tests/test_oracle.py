"""Anchor the oracle with the reference's analytic / property tests
(SURVEY.md S8c), re-expressed in NumPy.  CPU only."""
import numpy as np
import pytest

from oracle import mellon_oracle as mo


def _data(n=100, d=2, seed=0):
    rng = np.random.default_rng(seed)
    L = np.array([[2.0, 0.0], [1.0, 1.0]]) if d == 2 else np.eye(d)
    return rng.normal(size=(n, d)) @ L.T


def test_distance_quirk():
    # util.py:362-366: +1e-12 inside sqrt => dist(x,x) = 1e-6 exactly in spirit
    x = np.array([[1.0, 2.0], [3.0, 4.0]])
    D = mo.distance(x, x)
    assert np.allclose(np.diag(D), 1e-6, rtol=1e-3)
    assert np.isclose(D[0, 1], np.sqrt(8.0 + 1e-12))


def test_nn_distances_exact():
    # tests/test_parameters.py:244-258
    x = np.array([[1, 2], [2, 3], [3, 4]], dtype=float)
    assert np.allclose(mo.exact_nn_distances(x), np.sqrt(2))
    x = np.array([[1, 1], [2, 2], [4, 4], [5, 5]], dtype=float)
    assert np.allclose(mo.exact_nn_distances(x), np.sqrt(2))


def test_gp_type_table():
    # tests/test_parameters.py:271-290
    f = mo.compute_gp_type
    assert f(0, 100, 100) == mo.FULL
    assert f(100, 1.0, 100) == mo.FULL
    assert f(100, None, 100) == mo.FULL
    assert f(100, 0, 100) == mo.FULL
    assert f(100, 50, 100) == mo.FULL_NYSTROEM
    assert f(100, 0.5, 100) == mo.FULL_NYSTROEM
    assert f(50, 50, 100) == mo.SPARSE_CHOLESKY
    assert f(50, 1.0, 100) == mo.SPARSE_CHOLESKY
    assert f(50, None, 100) == mo.SPARSE_CHOLESKY
    assert f(50, 0, 100) == mo.SPARSE_CHOLESKY
    assert f(50, 25, 100) == mo.SPARSE_NYSTROEM
    assert f(50, 0.5, 100) == mo.SPARSE_NYSTROEM


def test_rank_and_n_landmarks_tables():
    # tests/test_parameters.py:318-365
    assert mo.compute_rank(mo.FULL_NYSTROEM) == 0.99
    assert mo.compute_rank(mo.SPARSE_CHOLESKY) == 1.0
    assert mo.compute_rank(None) == 1.0
    assert mo.compute_n_landmarks(None, 100, np.ones((50, 2))) == 50
    assert mo.compute_n_landmarks(None, 100, None) == 100
    assert mo.compute_n_landmarks(mo.FULL, 100, None) == 100
    assert mo.compute_n_landmarks(mo.SPARSE_CHOLESKY, 80, None) == 5000


@pytest.mark.parametrize("cls,form", [
    (mo.Matern32, lambda r: (1 + np.sqrt(3) * r) * np.exp(-np.sqrt(3) * r)),
    (mo.Matern52, lambda r: (1 + np.sqrt(5) * r + 5 * r * r / 3) * np.exp(-np.sqrt(5) * r)),
    (mo.ExpQuad, lambda r: np.exp(-r * r / 2)),
    (mo.Exponential, lambda r: np.exp(-r / 2)),
])
def test_kernel_closed_forms(cls, form):
    # tests/test_cov.py:41-64 evaluates at x=1, y in {2, 1.5}, ls=1.2
    x = np.array([[1.0]])
    for yv in (2.0, 1.5):
        y = np.array([[yv]])
        r = np.sqrt((1.0 - yv) ** 2 + 1e-12) / 1.2
        assert np.isclose(cls(1.2)(x, y)[0, 0], form(r), rtol=1e-13)


def test_ratquad_and_linear():
    x, y = _data(5), _data(4, seed=1)
    r = mo.distance(x, y) / 0.7
    assert np.allclose(mo.RatQuad(2.0, 0.7)(x, y), (r * r / 4.0 + 1) ** -2.0)
    assert np.allclose(mo.Linear(0.7)(x, y), x @ y.T / 0.7)


def test_active_dims_and_algebra():
    # tests/test_base_cov.py: Add/Mul/Pow, scalar right operand, nested active_dims
    x, y = _data(6, 2), _data(5, 2, seed=3)
    a, b = mo.Matern52(1.3, active_dims=0), mo.ExpQuad(0.9, active_dims=[1])
    assert np.allclose((a * b)(x, y), a(x, y) * b(x, y))
    assert np.allclose((a + b)(x, y), a(x, y) + b(x, y))
    assert np.allclose((a + 2.0)(x, y), a(x, y) + 2.0)
    assert np.allclose((3.0 * a)(x, y), a(x, y) * 3.0)
    assert np.allclose((a ** 2)(x, y), a(x, y) ** 2)
    assert np.allclose(a(x, y), mo.Matern52(1.3)(x[:, :1], y[:, :1]))
    t = mo.compute_cov_func(mo.Matern52, 1.1, ls_time=0.5)
    assert np.allclose(t(x, y), mo.Matern52(1.1)(x[:, :-1], y[:, :-1]) * mo.Matern52(0.5)(x[:, -1:], y[:, -1:]))


def test_ridge_closed_form_matches_sklearn():
    # parameters.py:895-896
    from sklearn.linear_model import Ridge
    rng = np.random.default_rng(0)
    L = rng.normal(size=(200, 17))
    nn = rng.uniform(0.1, 1.0, size=200)
    z0 = mo.compute_initial_value(nn, 3, -5.0, L)
    ref = Ridge(fit_intercept=False).fit(L, mo.mle(nn, 3) + 5.0).coef_
    assert np.allclose(z0, ref, rtol=1e-9, atol=1e-11)


def test_loss_gradient_finite_difference():
    rng = np.random.default_rng(1)
    L = rng.normal(size=(60, 9)) * 0.3
    nn = rng.uniform(0.1, 1.0, size=60)
    V, Vdr = mo.nn_likelihood_constants(nn, 4)
    z = rng.normal(size=9) * 0.1
    f0, g = mo.loss_and_grad(z, L, -3.0, V, Vdr)
    for j in range(9):
        e = np.zeros(9)
        e[j] = 1e-6
        fp, _ = mo.loss_and_grad(z + e, L, -3.0, V, Vdr)
        fm, _ = mo.loss_and_grad(z - e, L, -3.0, V, Vdr)
        assert np.isclose((fp - fm) / 2e-6, g[j], rtol=1e-6, atol=1e-7)


def test_lbfgsb_quadratic():
    # tests/test_inference.py:59-72
    res = mo.minimize_lbfgsb(lambda z: (float(z @ z), 2 * z), np.ones(4))
    assert res.loss < 1e-10


def test_laplace_std_is_hessian_diag():
    # tests/test_laplace.py: std == 1/sqrt(diag H); finite-difference the gradient
    rng = np.random.default_rng(2)
    L = rng.normal(size=(40, 5)) * 0.3
    nn = rng.uniform(0.1, 1.0, size=40)
    V, Vdr = mo.nn_likelihood_constants(nn, 2)
    z = rng.normal(size=5) * 0.1
    std = mo.laplace_std(z, L, -2.0, V)
    for j in range(5):
        e = np.zeros(5)
        e[j] = 1e-5
        gp = mo.loss_and_grad(z + e, L, -2.0, V, Vdr)[1][j]
        gm = mo.loss_and_grad(z - e, L, -2.0, V, Vdr)[1][j]
        assert np.isclose(1 / np.sqrt((gp - gm) / 2e-5), std[j], rtol=1e-6)


def test_not_positive_definite_raises():
    # decomposition.py:116-122
    x = np.zeros((4, 2))
    with pytest.raises(ValueError, match="not positively definite"):
        mo.full_rank(x, mo.Matern52(1.0) * -1.0, jitter=1e-6)


def test_density_predict_equals_fit_predict_sparse():
    # tests/test_density_estimator.py:40-44 (rel err < 1e-5)
    x = _data(100)
    fit = mo.density_fit(x, n_landmarks=10)
    assert fit.gp_type == mo.SPARSE_CHOLESKY
    pred = fit.predict(x)
    rel = np.std(pred - fit.log_density_x) / np.std(fit.log_density_x)
    assert rel < 1e-5


def test_density_full_and_1d():
    # tests/test_density_estimator.py:257-269 (1-D input)
    x = _data(100)
    fit = mo.density_fit(x)
    assert fit.gp_type == mo.FULL and fit.log_density_x.shape == (100,)
    pred = fit.predict(x)
    assert np.std(pred - fit.log_density_x) / np.std(fit.log_density_x) < 1e-5
    f1 = mo.density_fit(x[:, 0])
    assert f1.log_density_x.shape == (100,)


def test_density_approximations_close_to_default():
    # tests/test_density_estimator.py:80-96
    x = _data(100)
    full = mo.density_fit(x).log_density_x
    sparse = mo.density_fit(x, n_landmarks=10).log_density_x
    assert np.std(full - sparse) / np.std(full) < 2e-1


def test_predictor_feature_mismatch():
    # base_predictor.py:215-220
    x = _data(50)
    fit = mo.density_fit(x, n_landmarks=10)
    with pytest.raises(ValueError):
        fit.predict(np.zeros((3, 5)))


def test_dimensionality_guard():
    # density_estimator.py:327-333
    with pytest.raises(ValueError):
        mo.density_fit(np.random.default_rng(0).normal(size=(60, 51)), n_landmarks=10)


def test_covariance_dict_round_trip():
    c = mo.Matern52(1.5, active_dims=slice(None, -1)) * mo.Matern52(0.4, active_dims=-1)
    state = {
        "type": "mellon.Covariance",
        "left_data": {"type": "mellon.Covariance",
                      "data": {"ls": 1.5, "active_dims": {"type": "slice", "data": ["None", -1, "None"]}},
                      "metadata": {"classname": "Matern52", "module_name": "mellon.cov"}},
        "right_data": {"type": "mellon.Covariance", "data": {"ls": 0.4, "active_dims": -1},
                       "metadata": {"classname": "Matern52", "module_name": "mellon.cov"}},
        "active_dims": "None",
        "metadata": {"classname": "Mul", "module_name": "mellon"},
    }
    c2 = mo.Covariance.from_dict(state)
    x, y = _data(7, 2), _data(3, 2, seed=5)
    assert np.allclose(c(x, y), c2(x, y))


def test_predictive_uncertainty_identities():
    """conditional.py:409-440: at the training cells of a full GP the posterior variance is
    k(x,x) - sum((L^-1 K)^2) = jitter-level; mean covariance is PSD; diag == diagonal of the full form."""
    x = _data(60)
    fit = mo.density_fit(x)
    std = mo.laplace_std(fit.pre_transformation, fit.L, fit.mu, mo.nn_likelihood_constants(fit.nn_distances, fit.d)[0])
    p = fit.predict.attach_uncertainty(fit.Lp, std)
    xq = _data(15, seed=9)
    for f in (p.covariance, p.mean_covariance, p.uncertainty):
        full = f(xq, diag=False)
        assert full.shape == (15, 15) and np.allclose(np.diag(full), f(xq), rtol=1e-9, atol=1e-12)
    assert np.all(p.covariance(x) < 1e-4) and np.all(p.covariance(xq) > -1e-9)
    assert np.all(np.linalg.eigvalsh(p.mean_covariance(xq, diag=False)) > -1e-10)


# ---- Nystroem rank reduction (decomposition.py:23-76,126-171,213-266) -----------------------------------
def test_eigendecomposition_rank_rule():
    A = np.diag([5.0, 3.0, 1.0, 0.5, 0.25, -1e-9, 0.0])
    for rank, want in [(3, [1.0, 3.0, 5.0]), (10, [0.25, 0.5, 1.0, 3.0, 5.0]),
                       (0.5, [5.0]),              # cumsum 5, 8, 9, 9.5, 9.75; target 4.875 -> index 0 -> bumped to 1
                       (0.9, [3.0, 5.0]),         # target 8.775 -> index 2
                       (1.0, [0.5, 1.0, 3.0, 5.0])]:   # target 9.75 -> index 4 (the quirk: one short of all)
        s, v = mo.eigendecomposition(A, rank)
        assert np.allclose(s, want), (rank, s)
        assert v.shape == (7, len(want))


def test_select_rank_mirrors_oracle_rule():
    from mellon_amd.decomposition import _select_rank
    rng = np.random.default_rng(0)
    for _ in range(20):
        s = np.sort(np.concatenate([rng.lognormal(size=12) * 10.0 ** rng.uniform(-8, 2, 12), -rng.random(2) * 1e-12]))
        for rank in (0.5, 0.9, 0.99, 0.99999, 1.0, 1, 5, 12, 100):
            want = mo.eigendecomposition(np.diag(s), rank)[0].shape[0]
            assert _select_rank(s, rank) == want


def test_modified_low_rank_equals_projected_cholesky_factor():
    """The identity the device path uses: Q V sqrt(S) (reference) == B U_p with B = C Lp^-T and (S, U) the
    eigenpairs of B^T B."""
    x = mo.gaussian_mixture(400, 4, 3)
    xu = x[np.random.default_rng(1).choice(400, 50, replace=False)]
    cov = mo.Matern52(ls=2.0)
    B = mo.standard_low_rank(x, cov, xu)
    S, U = np.linalg.eigh(B.T @ B)
    for rank in (0.99, 0.999, 10, 50):
        ref = mo.modified_low_rank(x, cov, xu, rank=rank)
        p = ref.shape[1]
        mine = B @ U[:, -p:]
        assert np.abs(ref @ ref.T - mine @ mine.T).max() < 1e-12
        assert np.abs(np.abs(ref) - np.abs(mine)).max() < 1e-10          # column-wise, up to sign
        assert np.allclose((ref * ref).sum(0), S[-p:], rtol=1e-9)


def test_density_fit_nystroem_types():
    rng = np.random.default_rng(0)
    x = rng.normal(size=(100, 2)) @ np.array([[2.0, 0.0], [1.0, 1.0]]).T
    base = mo.density_fit(x).log_density_x
    for kw, gp in [(dict(rank=0.99, n_landmarks=80), "sparse_nystroem"), (dict(rank=0.99, n_landmarks=0), "full_nystroem")]:
        f = mo.density_fit(x, **kw)
        assert f.gp_type == gp and f.Lp is None and f.L.shape[1] < 80
        assert np.std(f.predict(x) - base) / np.std(base) < 2e-1      # tests/test_density_estimator.py:80-96


# ---- analytic kernel gradients (tests/test_cov.py:17-64, tests/test_base_cov.py:41-53,87-99,132-144,173-185) ----
_ACTIVE_DIMS = [None, slice(2), 1, slice(None, None, 2), [1, 2]]


def _fd_k_grad(cov, x, y, h=1e-5):
    out = np.zeros((x.shape[0], y.shape[0], y.shape[1]))
    for k in range(y.shape[1]):
        e = np.zeros(y.shape[1])
        e[k] = h
        out[:, :, k] = (cov.k(x, y + e) - cov.k(x, y - e)) / (2 * h)
    return out


@pytest.mark.parametrize("name", ["Matern32", "Matern52", "ExpQuad", "Exponential", "RatQuad", "Linear"])
@pytest.mark.parametrize("active_dims", _ACTIVE_DIMS)
def test_k_grad_matches_numerical_derivative(name, active_dims):
    # the reference compares k_grad with jax.jacfwd at these very points (atol 1e-6); central differences
    # of the restated k() stand in for autodiff
    cls = getattr(mo, name)
    cov = cls(3, 1.2, active_dims=active_dims) if name == "RatQuad" else cls(1.2, active_dims=active_dims)
    x = np.ones((5, 4))
    y = np.ones((6, 4)) * 2
    y[1] = 1.5
    g = cov.k_grad(x)(y)
    assert g.shape == (5, 6, 4)
    assert np.allclose(g, _fd_k_grad(cov, x, y), atol=1e-6)


@pytest.mark.parametrize("active_dims", _ACTIVE_DIMS)
def test_k_grad_composites(active_dims):
    x = np.ones((2, 3))
    for cov in (mo.Add(mo.Matern32(1.4), mo.Exponential(3.4), active_dims=active_dims),
                mo.Mul(mo.Matern32(1.4), mo.Exponential(3.4), active_dims=active_dims),
                mo.Pow(mo.Matern32(1.4), 3.2, active_dims=active_dims)):
        assert np.allclose(cov.k_grad(x)(2 * x), _fd_k_grad(cov, x, 2 * x), atol=1e-6)
    h = 0.2 + 1.1 * mo.Matern52(1.4, active_dims=0) + \
        2.1 * mo.Exponential(3.4, active_dims=[1, 2]) * mo.RatQuad(1.1, 3.4, active_dims=slice(0, 2, 1)) + \
        mo.Matern52(1.0, active_dims=[False, True, True])
    y = 2 * x ** 2
    assert np.allclose(h.k_grad(x)(y), _fd_k_grad(h, x, y), atol=1e-6)


def test_predictor_gradient_restatement_is_consistent():
    rng = np.random.default_rng(0)
    c, w = rng.normal(size=(30, 3)), rng.normal(size=30)
    pred = mo.Predictor(mo.Matern52(1.3), c, w, -2.0, 30)
    xq = rng.normal(size=(7, 3))
    analytic = np.einsum("j,jik->ik", w, pred.cov_func.k_grad(c)(xq))    # symmetry: d k(x, c_j)/dx = k_grad(c)(x)[j]
    assert np.abs(pred.gradient(xq) - analytic).max() < 1e-8


# --- per-feature ("per-gene") and per-cell noise: the reference's own properties (tests/test_pergene_sigma.py) ---------
def _multi_output_data(n=50, d=2, p=3, seed=42):
    rng = np.random.default_rng(seed)
    return rng.normal(size=(n, d)), rng.normal(size=(n, p)), np.array([0.5, 1.0, 2.0])[:p]


@pytest.mark.parametrize("n_landmarks", [0, 15])
def test_oracle_pergene_matches_per_column_scalar(n_landmarks):
    """tests/test_pergene_sigma.py:34-132: predictions, leverage, obs_variance and loo residuals of one fit with a
    sigma per output equal the per-column scalar fits at atol 1e-5."""
    X, Y, sigma = _multi_output_data()
    pg = mo.function_fit(X, Y, sigma, n_landmarks=n_landmarks, obs_variance=True)
    assert pg.leverage(X).shape == Y.shape
    for g in range(Y.shape[1]):
        sc = mo.function_fit(X, Y[:, g], float(sigma[g]), n_landmarks=n_landmarks, obs_variance=True)
        np.testing.assert_allclose(pg(X)[:, g], sc(X), atol=1e-5)
        np.testing.assert_allclose(pg.leverage(X)[:, g], sc.leverage(X), atol=1e-5)
        np.testing.assert_allclose(pg.obs_variance(X)[:, g], sc.obs_variance(X), atol=1e-5)
        np.testing.assert_allclose(pg.loo_residuals_squared(X, Y)[:, g], sc.loo_residuals_squared(X, Y[:, g]), atol=1e-5)
    lev = pg.leverage(X)
    assert np.all(lev >= 0) and np.all(lev < 1)                     # test_pergene_sigma.py:125-135


def test_oracle_sigma_shape_rules():
    """tests/test_pergene_sigma.py:160-220: (p,) with n == p is per-feature, (1, p) == (p,), (n, 1) is not."""
    y = np.ones((30, 3))
    assert mo.is_per_feature_sigma(np.ones(3), y) and mo.is_per_feature_sigma(np.ones((1, 3)), y)
    assert mo.is_per_feature_sigma(np.ones((30, 3)), y)
    assert not mo.is_per_feature_sigma(np.ones((30, 1)), y) and not mo.is_per_feature_sigma(0.5, y)
    assert mo.is_per_feature_sigma(np.ones(20), np.ones((20, 20)))
    X, Y, sigma = _multi_output_data(30)
    a = mo.function_fit(X, Y, sigma[None, :], n_landmarks=0)(X)
    b = mo.function_fit(X, Y, sigma, n_landmarks=0)(X)
    np.testing.assert_allclose(a, b, atol=1e-10)


@pytest.mark.parametrize("n_landmarks", [0, 15])
def test_oracle_per_cell_sigma_constant_equals_scalar(n_landmarks):
    """An element-wise sigma vector that happens to be constant is the scalar model (conditional.py:155-159)."""
    X, Y, _ = _multi_output_data()
    a = mo.function_fit(X, Y[:, 0], np.full(50, 0.7), n_landmarks=n_landmarks)(X)
    b = mo.function_fit(X, Y[:, 0], 0.7, n_landmarks=n_landmarks)(X)
    np.testing.assert_allclose(a, b, atol=1e-9)


def test_oracle_hessian_is_the_derivative_of_the_fd_gradient():
    """The oracle Hessian (FD of the analytic k_grad contraction) agrees with second differences of the mean itself."""
    rng = np.random.default_rng(4)
    c, w = rng.normal(size=(30, 3)), rng.normal(size=30)
    pred = mo.Predictor(mo.Matern52(ls=1.7), c, w, 0.3, 30)
    X = rng.normal(size=(6, 3))
    H = pred.hessian(X)
    assert H.shape == (6, 3, 3) and np.allclose(H, np.swapaxes(H, 1, 2))
    h = 1e-3
    for a in range(3):
        for b in range(3):
            ea, eb = np.eye(3)[a] * h, np.eye(3)[b] * h
            fd = (pred(X + ea + eb) - pred(X + ea - eb) - pred(X - ea + eb) + pred(X - ea - eb)) / (4 * h * h)
            assert np.abs(fd - H[:, a, b]).max() < 1e-5 * max(np.abs(H).max(), 1.0)
    s, ld = pred.hessian_log_determinant(X)
    assert s.shape == ld.shape == (6,)
