"""Round 5 (-m gpu): the preconditioner factored in w-space, the deferred landmark factor and its batched factorisation,
the GEMM's batch dimension.  Everything goes through libmellon_hip.so; the oracle is the checker."""
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
    return mo.Covariance.from_dict(product_cov.to_dict())


def relmax(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _problem(n, d, m, seed):
    x = mo.gaussian_mixture(n, d, seed=seed)
    nn = mo.exact_nn_distances(x)
    ls = mo.compute_ls(nn)
    mu = mo.compute_mu(nn, d)
    rng = np.random.default_rng(seed)
    xu = x[np.sort(rng.choice(n, m, replace=False))]
    return x, nn, ls, mu, xu


@pytest.mark.parametrize("n,d,m,stride", [(4000, 6, 300, 1), (9000, 10, 517, 4), (3000, 4, 129, 3)])
def test_implicit_fit_deferred_lp_and_w_space_preconditioner(ctx, n, d, m, stride):
    """Implicit fits defer Lp = chol(cov(xu, xu) + jitter I) and factor it in ONE chain of launches with the preconditioner's
    matrix M = s K_s^T K_s + Kj (dev_cholesky_lower2).  Checked here: the deferred factor equals the oracle's
    (decomposition.py:111-123); the Ridge start equals the oracle's on the same cells (parameters.py:895-896); the variable
    change z = C^-T u with C = Lp^-1 R round-trips and whitens the Ridge matrix; and the preconditioned objective is the
    reference's objective of z (inference.py:167-192)."""
    from mellon_amd import cov
    x, nn, ls, mu, xu = _problem(n, d, m, seed=n + m)
    c = cov.Matern52(ls)
    oc = _pair(c)
    fit = ctx.fit_prepare(c.lower(d), x, xu, 1e-6, implicit=True)
    V, Vdr = mo.nn_likelihood_constants(nn, d)
    fit.set_likelihood(V, Vdr, mu)
    fit.precond_build(stride, 0, force=True)                 # batched: chol(M) and chol(Kj) side by side
    Lp_ref = mo.full_rank(xu, oc)
    assert relmax(fit.Lp(), Lp_ref) < 1e-8
    L_ref = mo.standard_low_rank(x, oc, xu, Lp=Lp_ref)
    # round trip of the variable change and the defining property C C^T = I + s L_s^T L_s  (mode 0: u = C^T z; 2: g_u = C^-1 g_z)
    rng = np.random.default_rng(1)
    z = rng.normal(size=m)
    u = fit.precond_apply(0, z)
    assert relmax(fit.precond_apply(1, u), z) < 1e-9
    Ls = L_ref[::stride]
    H = np.eye(m) + stride * (Ls.T @ Ls)
    # C^-1 H C^-T = I  <=>  C^-1 (H (C^-T v)) = v
    v = rng.normal(size=m)
    zz = fit.precond_apply(1, v)                             # C^-T v
    back = fit.precond_apply(2, H @ zz)                      # C^-1 H C^-T v
    # (stride 1: the exact fp64 Gram -- what is left is eps * |M| / lambda_min(M), M = K^T K + Kj with lambda_min ~ the
    #  jitter: the same bound the whitened form had through |Lp^-1|^2; a sampled Gram is the 23-bit integer one)
    tol = 1e-4 if stride == 1 else 2e-3
    assert relmax(back, v) < tol, relmax(back, v)
    # the preconditioned objective is the reference's objective at z = C^-T u, its gradient C^-1 grad_z
    loss_u, grad_u, z_of_u = fit.objective_precond(u)
    loss_ref, grad_ref = mo.loss_and_grad(z, L_ref, mu, V, Vdr)
    assert relmax(z_of_u, z) < 1e-9
    assert abs(loss_u - loss_ref) / abs(loss_ref) < 1e-11
    assert relmax(grad_u, fit.precond_apply(2, grad_ref)) < 1e-8
    if stride == 1:
        z0 = fit.ridge_init(mo.mle(nn, d) - mu)
        assert relmax(z0, mo.compute_initial_value(nn, d, mu, L_ref)) < 1e-5   # (a start value: same conditioning remark)


def test_deferred_lp_not_positive_definite_raises_the_reference_error(ctx):
    """A deferred landmark factor that fails surfaces as the reference's ValueError (decomposition.py:116-122) with the FIT's
    jitter in the message, from whichever call factors it: the batched chain of the preconditioner, or a direct request."""
    from mellon_amd import cov
    rng = np.random.default_rng(0)
    x = rng.normal(size=(400, 3))
    bad = (cov.Matern52(1.0) * -1.0).lower(3)
    fit = ctx.fit_prepare(bad, x, x[:64], 1e-6, implicit=True)          # deferred: nothing factored yet
    with pytest.raises(ValueError, match=r"not positively definite with jitter=1e-06"):
        fit.precond_build(1, 0, force=True)
    fit2 = ctx.fit_prepare(bad, x, x[:64], 1e-6, implicit=True)
    with pytest.raises(ValueError, match=r"not positively definite with jitter=1e-06"):
        fit2.Lp()
    # the explicit route factors inside fit_prepare, as before
    with pytest.raises(ValueError, match="not positively definite"):
        ctx.fit_prepare(bad, x, x[:64], 1e-6)


def test_estimator_default_is_pure_fp64_and_matches_the_oracle(ctx, monkeypatch):
    """Round 5: the product default is the pure-fp64 solve (the 32-bit copy is opt-in, MELLON_AMD_MIXED=1); the default call
    reproduces the oracle's tight optimum to the 1e-5 of BASELINE.json, Lp and the pre-transformation included."""
    import mellon_amd as mellon
    monkeypatch.delenv("MELLON_AMD_MIXED", raising=False)
    monkeypatch.setenv("MELLON_AMD_MIXED_MIN_ELEMS", "0")                # (would force the copy if mixed were still the default)
    x = mo.gaussian_mixture(8000, 8, seed=11)
    nn = mo.exact_nn_distances(x)
    rng = np.random.default_rng(3)
    lm = x[np.sort(rng.choice(8000, 400, replace=False))]
    ref = mo.density_fit(x, landmarks=lm, nn_distances=nn, lbfgsb_options=mo.LBFGSB_TIGHT)
    est = mellon.DensityEstimator(landmarks=lm, nn_distances=nn)
    dens = est.fit_predict(x)
    st = est._fit.stage_times()
    assert st["objective32_launches"] == 0 and st["objective_launches"] > 0
    scale = np.abs(ref.log_density_x).max()
    assert np.abs(dens - ref.log_density_x).max() / scale < 1e-5
    assert np.std(dens - ref.log_density_x) / np.std(ref.log_density_x) < 1e-5
    assert relmax(np.asarray(est.Lp), ref.Lp) < 1e-8
    z = np.asarray(est.pre_transformation)
    assert relmax(z, ref.pre_transformation) < 1e-4                      # (z carries cond(Lp); the density above is the contract)
    # predict(X) == fit_predict(X)  (tests/test_density_estimator.py:40-44)
    assert np.abs(est.predict(x[:500]) - dens[:500]).max() / scale < 1e-9


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0)])
def test_gemm_batch_dimension(ctx, ta, tb):
    """GemmArgs.batch: two independent products in one launch (grid z) equal the two single launches bit for bit."""
    r = ctx.diag_dgemm_batch(ta, tb, 700, 333, 256)
    assert r == 0.0, r


def _tree(n, d, seed, branches=6):
    """Diffusion-map-like coordinates (tools/hard_cases.py): cells along a branching tree of smooth curves in a 3-D latent
    space, embedded by a random smooth map, column k scaled by 0.8^k, unevenly populated."""
    rng = np.random.default_rng(seed)
    t = rng.beta(0.7, 1.3, size=n)
    b = rng.integers(0, branches, size=n)
    dirs = rng.normal(size=(branches, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    bend = rng.normal(size=(branches, 3)) * 0.5
    z = t[:, None] * dirs[b] + (t ** 2)[:, None] * bend[b] + 0.02 * (1 + 3 * t)[:, None] * rng.normal(size=(n, 3))
    W1 = rng.normal(size=(3, d)); W2 = rng.normal(size=(3, d))
    x = np.tanh(z @ W1) + 0.3 * np.sin(2.0 * z @ W2)
    return np.ascontiguousarray(x * (0.8 ** np.arange(d))[None, :])


@pytest.mark.parametrize("case", ["tree", "heavy tails"])
def test_hard_data_against_the_oracle(case, monkeypatch):
    """The data the solver's shortcuts were NOT tuned on, at a size the oracle finishes: the default solve (capped start,
    subsample phase, repeated rebuilds) against oracle.density_fit at its tight stopping rule -- 1e-5 on the log-density
    (BASELINE.json) -- and against the reference AS RUN (SciPy L-BFGS-B at its defaults, inference.py:272-288), which stops
    at its 500-iteration limit or its ftol long before: the device solve must be at least as well converged."""
    import mellon_amd as mellon
    monkeypatch.delenv("MELLON_AMD_MIXED", raising=False)
    n, d, m = 20_000, 20, 500
    x = _tree(n, d, 5) if case == "tree" else np.random.default_rng(6).standard_t(3, size=(n, d))
    nn = mo.exact_nn_distances(x)
    lm = x[np.sort(np.random.default_rng(7).choice(n, m, replace=False))]
    ref = mo.density_fit(x, landmarks=lm, nn_distances=nn, lbfgsb_options=mo.LBFGSB_TIGHT)
    as_run = mo.density_fit(x, landmarks=lm, nn_distances=nn)                  # the reference's own stopping rule
    est = mellon.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
    dens = est.fit_predict(x)
    assert est.opt_state.success
    scale = np.abs(ref.log_density_x).max()
    err = np.abs(dens - ref.log_density_x).max() / scale
    err_as_run = np.abs(as_run.log_density_x - ref.log_density_x).max() / scale
    assert err < 1e-5, (case, err)
    assert np.std(dens - ref.log_density_x) / np.std(ref.log_density_x) < 1e-5
    assert err <= max(err_as_run, 1e-7), (case, err, err_as_run)
    st = est._fit.stage_times()
    print(f"{case}: device {est.loss_func.n_eval} evaluations / {st['objective_pass_equivalents']:.1f} pass-equivalents, "
          f"{int(st['precond_rebuilds'])} rebuilds, {err:.1e} from the tight optimum; reference as run: {err_as_run:.1e}")


def test_full_conditional_with_y_cov_factor(ctx):
    """`y_cov_factor=`: the caller's own left factor of the observation noise (conditional.py:69-81,101-135,253-306) --
    L = chol(K + M M^T [+ diagonal up to the jitter]), weights = L^-T L^-1 (y - mu), W = L^-T L^-1 M -- against the oracle:
    mean, covariance and mean covariance of the predictor."""
    from mellon_amd import cov
    from mellon_amd.conditional import FullConditional
    rng = np.random.default_rng(8)
    n, d = 700, 3
    x = rng.normal(size=(n, d))
    y = np.sin(x @ rng.normal(size=d)) + 0.1 * rng.normal(size=n)
    M = np.concatenate([0.3 * rng.normal(size=(n, 2)), np.diag(np.full(n, 0.05))[:, :5]], axis=1)   # n x 7
    M[n // 2:, :] *= 1e-5                                   # rows whose noise variance is under the jitter: the diagonal correction acts
    c = cov.Matern52(1.3)
    oc = _pair(c)
    got = FullConditional(x, y, 0.2, c, y_cov_factor=M, sigma=None, with_uncertainty=True)
    ref = mo.full_conditional(x, y, 0.2, oc, sigma=None, y_cov_factor=M, with_uncertainty=True)
    xq = rng.normal(size=(300, d))
    assert relmax(got.mean(xq), ref.mean(xq)) < 1e-7
    assert relmax(got.covariance(xq), ref.covariance(xq)) < 1e-6
    assert relmax(got.mean_covariance(xq), ref.mean_covariance(xq)) < 1e-6
    with pytest.raises(ValueError, match="either `sigma` or `y_cov_factor`"):
        FullConditional(x, y, 0.2, c, y_cov_factor=M, sigma=0.3)


@pytest.mark.parametrize("n,d,m", [(20_000, 10, 300), (5000, 3, 16), (60_000, 50, 700)])
def test_kmeans_sklearn_compatible_seeding(ctx, n, d, m):
    """init="sklearn": the device picks the cells sklearn's own k-means++ picks (sklearn.cluster.kmeans_plusplus with
    RandomState(42): the seeding behind the reference's k_means(x, m, n_init=1, random_state=42), parameters.py:275-291) --
    every one of the m seeds, not only the first 16 -- and Lloyd's sweeps from them end at sklearn's clustering quality."""
    from sklearn.cluster import kmeans_plusplus, k_means
    x = mo.gaussian_mixture(n, d, seed=21)
    _, idx_ref = kmeans_plusplus(x, m, random_state=42)
    seeds = ctx.kmeans(x, m, seed=42, max_iter=0, init="sklearn")
    idx = ctx.last_kmeans_seed_indices
    assert np.array_equal(idx[:16], idx_ref[:16])
    assert np.array_equal(idx, idx_ref), int(np.argmax(idx != idx_ref))
    assert np.array_equal(seeds, x[idx_ref])
    if n <= 20_000:
        c_ref, _, inertia_ref = k_means(x, m, n_init=1, random_state=42)
        c, _, inertia = ctx.kmeans(x, m, seed=42, init="sklearn", return_info=True)
        assert abs(inertia - inertia_ref) <= 2e-3 * inertia_ref
        # most centres coincide (a cell tied between two centres within the fp16 pre-filter's 1e-5 may go either way)
        dist = np.sqrt(((c[:, None, :] - c_ref[None, :, :]) ** 2).sum(-1)).min(axis=1)
        assert np.median(dist) < 1e-6 * np.abs(x).max()


def test_one_upload_for_the_steps_before_the_fit(ctx):
    """DensityEstimator.prepare_inference uploads large host cells ONCE for the 1-NN search, the k-means landmarks and the fit
    (reference density_estimator.py:404-470 hands the same host array to each step): the landmarks, distances and densities
    are those of the steps run one by one from the host array, and the HBM copy is gone afterwards."""
    import mellon_amd
    from mellon_amd import _lib
    from mellon_amd.parameters import compute_landmarks, compute_nn_distances
    n, d, m = 180_000, 50, 1500                    # n m > KMEANS_DEVICE_THRESHOLD, 72 MB of cells
    x = mo.gaussian_mixture(n, d, seed=5)
    est = mellon_amd.DensityEstimator(n_landmarks=m, check_rank=False)
    est.set_x(x)
    dev = est._x_on_device()
    assert isinstance(dev, _lib.DeviceArray) and est._x_on_device() is dev and est._x_for_fit() is dev
    est._release_x_on_device()
    assert est._x_for_fit() is est.x
    est = mellon_amd.DensityEstimator(n_landmarks=m, check_rank=False)
    dens = est.fit_predict(x)
    assert "_x_dev" not in est.__dict__            # released when prepare_inference() is through
    lm = compute_landmarks(x, n_landmarks=m, random_state=est.random_state)
    nn = compute_nn_distances(x)
    assert np.array_equal(np.asarray(est.landmarks), lm)
    assert np.array_equal(np.asarray(est.nn_distances), nn)
    ref = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False).fit_predict(x)
    assert relmax(dens, ref) < 1e-9
    # small inputs are handed to each step as they are
    small = mellon_amd.DensityEstimator(n_landmarks=50, check_rank=False)
    small.set_x(x[:2000])
    assert small._x_on_device() is small.x
