"""BASELINE.json configs C3, C4, C5 at FULL size on the GPU: the size-independent properties the domain offers
(predict(X) == fit_predict(X); first-order optimality of the MAP solution; the sharded fit == the unsharded one;
column-wise consistency of the batched function estimator), plus C2 against the oracle at full size.  The oracle
itself cannot finish C3-C5 in test time (minutes to hours, >= 80 GB at C3), so these complement the oracle-checked
cases at reduced size in test_gpu_estimators.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def gaussian_mixture(n, d, seed, k=10):
    """BASELINE.md S2 synthetic cells (same generator as bench.py)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    means = rng.normal(0.0, 3.0, size=(k, d))
    sig = rng.uniform(0.5, 1.5, size=k)
    comp = rng.integers(0, k, size=n)
    x = means[comp] + rng.normal(size=(n, d)) * sig[comp][:, None]
    return np.ascontiguousarray(x[rng.permutation(n)])


def relmax(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.fixture(scope="module")
def ctx():
    from mellon_amd import _lib
    return _lib.default_context()


@pytest.fixture(scope="module")
def c3(ctx):
    n, d, m = 1_000_000, 50, 5000
    x = gaussian_mixture(n, d, 3)
    lm = ctx.kmeans(x[:100_000], m, seed=42)
    lm = np.ascontiguousarray(lm.astype(np.float32).astype(np.float64))
    nn = ctx.nn_distances(x)
    return x, lm, nn


def test_c3_full_size_properties(ctx, c3):
    """C3: 1e6 cells x 50 dims, 5000 landmarks, Matern52 (the bench workload)."""
    import mellon_amd
    x, lm, nn = c3
    est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
    dens = est.fit_predict(x)
    assert dens.shape == (x.shape[0],) and np.all(np.isfinite(dens))
    # predict(X) == fit_predict(X) (the reference's own property, tests/test_density_estimator.py:40-44, rel 1e-5)
    k = 50_000
    assert relmax(est.predict(x[:k]), dens[:k]) < 1e-9
    assert relmax(est.predict(x[-k:]), dens[-k:]) < 1e-9
    # first-order optimality of the strictly convex MAP objective at the returned pre_transformation: the gradient in
    # the preconditioned variable (where the Hessian is ~identity) is tiny compared with its size at the Ridge start
    lf = est.loss_func
    _, g_opt = lf.value_and_grad_u(lf.u_from_z(est.pre_transformation))
    _, g_start = lf.value_and_grad_u(lf.u_from_z(est.initial_value))
    assert np.abs(g_opt).max() < 1e-8 * np.abs(g_start).max()
    # the product default is the pure-fp64 solve (round 5; the 32-bit copy is opt-in)
    stats = est._fit.stage_times()
    assert stats["objective32_launches"] == 0 and stats["objective_launches"] > 0
    est._fit.close()


def test_c3_full_size_fp64_only_and_sharded(ctx, c3, monkeypatch):
    """The same fit (i) with the opt-in 32-bit copy (MELLON_AMD_MIXED=1: warm-up passes at half the bytes, the optimum
    that of the fp64 objective) and (ii) cell-sharded over four thread-ranks with real collectives: same log-density."""
    import mellon_amd
    from mellon_amd import distributed
    x, lm, nn = c3
    monkeypatch.delenv("MELLON_AMD_MIXED", raising=False)
    est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
    dens = est.fit_predict(x)
    assert est._fit.stage_times()["objective32_launches"] == 0
    est._fit.close()
    monkeypatch.setenv("MELLON_AMD_MIXED", "1")
    est32 = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
    dens32 = est32.fit_predict(x)
    assert est32._fit.stage_times()["objective32_launches"] > 0
    est32._fit.close()
    monkeypatch.delenv("MELLON_AMD_MIXED")
    assert relmax(dens32, dens) < 1e-6

    def body(comm):
        lo, hi = distributed.shard_bounds(x.shape[0], comm.world_size, comm.rank)
        e = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn[lo:hi], check_rank=False)
        out = e.fit_predict(np.ascontiguousarray(x[lo:hi]))
        e._fit.close()
        return out

    sharded = np.concatenate(distributed.run_loopback(4, body))
    assert relmax(sharded, dens) < 1e-6


def test_c2_full_size_against_oracle(ctx):
    """C2: 1e5 cells x 20 dims, 1000 landmarks, ExpQuad -- the largest config the oracle finishes in test time."""
    import mellon_amd
    from oracle import mellon_oracle as mo
    n, d, m = 100_000, 20, 1000
    x = gaussian_mixture(n, d, 2)
    lm = np.ascontiguousarray(ctx.kmeans(x[:50_000], m, seed=42).astype(np.float32).astype(np.float64))
    nn = ctx.nn_distances(x)
    est = mellon_amd.DensityEstimator(cov_func_curry=mellon_amd.cov.ExpQuad, landmarks=lm, nn_distances=nn,
                                      check_rank=False)
    dens = est.fit_predict(x)
    ref = mo.density_fit(x, cov_func_curry=mo.ExpQuad, landmarks=lm, nn_distances=nn, lbfgsb_options=mo.LBFGSB_TIGHT)
    assert abs(est.mu - ref.mu) < 1e-10 and abs(est.ls - ref.ls) < 1e-10 * ref.ls
    assert relmax(dens, ref.log_density_x) < 1e-5
    assert np.std(dens - ref.log_density_x) / np.std(ref.log_density_x) < 1e-5
    assert relmax(est.predict(x[:20_000]), ref.log_density_x[:20_000]) < 1e-5


def test_c4_full_size_properties(ctx):
    """C4: TimeSensitiveDensityEstimator, 5e5 cells x 30 dims + time, 8 time points, 2000 landmarks, ls_time 1.5."""
    import mellon_amd
    n, d, m, T = 500_000, 30, 2000, 8
    xs = gaussian_mixture(n, d, 4)
    times = np.repeat(np.arange(float(T)), n // T)
    xs = xs + 0.2 * times[:, None]                      # component means drift linearly in time
    xt = np.ascontiguousarray(np.concatenate([xs, times[:, None]], axis=1))
    nn = np.empty(n)
    for t in range(T):
        idx = np.flatnonzero(times == t)
        nn[idx] = ctx.nn_distances(np.ascontiguousarray(xs[idx]))
    ls = float(np.exp(np.log(nn).mean() + 3.0))
    rng = np.random.default_rng(4)
    sub = xt[rng.choice(n, 50_000, replace=False)].copy()
    sub[:, -1] *= ls / 1.5                              # parameters.py:294-349: k-means with the time column rescaled
    lm = ctx.kmeans(sub, m, seed=42)
    lm[:, -1] /= ls / 1.5
    est = mellon_amd.TimeSensitiveDensityEstimator(landmarks=lm, nn_distances=nn, ls_time=1.5, d=d, check_rank=False)
    dens = est.fit_predict(xt)
    assert np.all(np.isfinite(dens)) and abs(est.ls - ls) < 1e-10 * ls
    assert "Mul" in type(est.cov_func).__name__
    k = 40_000
    assert relmax(est.predict(xt[:k]), dens[:k]) < 1e-9
    assert relmax(est.predict(xs[-k:], times[-k:]), dens[-k:]) < 1e-9
    lf = est.loss_func
    _, g_opt = lf.value_and_grad_u(lf.u_from_z(est.pre_transformation))
    _, g_start = lf.value_and_grad_u(lf.u_from_z(est.initial_value))
    assert np.abs(g_opt).max() < 1e-8 * np.abs(g_start).max()


def test_c5_full_size_properties(ctx):
    """C5: FunctionEstimator, 2e5 cells x 50 dims, 2000 outputs, sigma 0.1, 2000 landmarks, batched predict."""
    import mellon_amd
    n, d, m, p = 200_000, 50, 2000, 2000
    rng = np.random.default_rng(5)
    x = gaussian_mixture(n, d, 5)
    W = rng.normal(size=(d, p)) / np.sqrt(d)
    y = np.sin(x @ W) + 0.1 * rng.normal(size=(n, p))
    lm = ctx.kmeans(x[:100_000], m, seed=42)
    nn = ctx.nn_distances(x)
    est = mellon_amd.FunctionEstimator(sigma=0.1, landmarks=lm, nn_distances=nn)
    est.fit(x, y)
    k = 20_000
    pred = est.predict(x[:k])
    assert pred.shape == (k, p) and np.all(np.isfinite(pred))
    # column-wise consistency: output j of the batched solve == a single-output fit on column j
    for j in (0, 7, p - 1):
        one = mellon_amd.FunctionEstimator(sigma=0.1, landmarks=lm, nn_distances=nn).fit(x, y[:, j]).predict(x[:k])
        assert relmax(pred[:, j], one) < 1e-9
    # linearity of the conditional mean in y (the weights solve is linear): fit(a y1 + b y2) == a fit(y1) + b fit(y2)
    comb = 0.3 * y[:, 1] - 1.7 * y[:, 2]
    lin = mellon_amd.FunctionEstimator(sigma=0.1, landmarks=lm, nn_distances=nn).fit(x, comb).predict(x[:k])
    assert relmax(lin, 0.3 * pred[:, 1] - 1.7 * pred[:, 2]) < 1e-8
    # the smoother recovers the signal: residual close to the injected noise level
    assert 0.05 < float(np.std(pred - y[:k])) < 0.5
