"""Round 6 (-m gpu): oracle anchors where the solver is hardest, at the largest size the oracle finishes in test time
(C2-sized: 1e5 cells x 20 dims, 1000 landmarks); the advisor's round-5 findings.  Everything goes through
libmellon_hip.so; the oracle is the checker."""
import copy
import os
import pickle

import numpy as np
import pytest

from oracle import mellon_oracle as mo
from test_gpu_round5 import _tree

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from mellon_amd import _lib
    return _lib.default_context()


def relmax(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("case", ["tree", "heavy tails"])
def test_hard_data_at_c2_size_against_the_oracle(ctx, case, monkeypatch):
    """tools/robustness_sweep_large.py's two hardest generators (a diffusion-map-like tree in 20 dimensions; Student-t3
    tails) at BASELINE config 2's size against oracle.density_fit at its TIGHT stopping rule (inference.py:35-92,272-288
    restated; the optimum is unique by strict convexity).  The oracle's L-BFGS-B is STARTED at the device's
    pre_transformation -- a choice of starting point only: what certifies its result is a Newton step of the oracle's OWN
    objective at its own final point (oracle gradient, closed-form Hessian), which must move the log-density by < 1e-6 of
    its scale.  A wrong device optimum would send the oracle away from it and fail the comparison.
    Gates: 5e-6 on the log-density (BASELINE.json's 1e-5 with a factor of two in hand), and default path == plain path (no
    subsample phase, no rebuilds) to 5e-6."""
    import mellon_amd as mellon
    monkeypatch.delenv("MELLON_AMD_MIXED", raising=False)
    n, d, m = 100_000, 20, 1000
    x = _tree(n, d, 15) if case == "tree" else np.random.default_rng(16).standard_t(3, size=(n, d))
    x = np.ascontiguousarray(x)
    nn = ctx.nn_distances(x)
    lm = np.ascontiguousarray(ctx.kmeans(x[:50_000], m, seed=42).astype(np.float32).astype(np.float64))
    est = mellon.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
    dens = est.fit_predict(x)
    assert est.opt_state.success
    st = est._fit.stage_times()
    ref = mo.density_fit(x, landmarks=lm, nn_distances=nn, lbfgsb_options=mo.LBFGSB_TIGHT,
                         initial_value=np.asarray(est.pre_transformation, dtype=np.float64))
    assert abs(est.mu - ref.mu) < 1e-10 and abs(est.ls - ref.ls) < 1e-10 * ref.ls
    # Certificate of the oracle's point, independent of anybody's stopping rule: one Newton step of the oracle's objective
    # (inference.py:35-92: grad = z - L^T (1 - a), Hessian = I + L^T diag(a) L, a = e^{f + V}; the gradient is the oracle's
    # own) measures the distance to the unique optimum to second order.  (L-BFGS-B in the un-preconditioned z stops on its
    # relative-decrease test with |g|_2 ~ 1e-3 on these data sets: the loss is ~1e6 and its rounding noise hides the rest.)
    V, Vdr = mo.nn_likelihood_constants(mo.validate_nn_distances(nn), d)
    _, g = mo.loss_and_grad(ref.pre_transformation, ref.L, ref.mu, V, Vdr)
    a = np.exp(ref.log_density_x + V)
    H = ref.L.T @ (a[:, None] * ref.L)
    H[np.diag_indices_from(H)] += 1.0
    dz = np.linalg.solve(H, g)
    df = ref.L @ dz
    scale = np.abs(ref.log_density_x).max()
    assert np.abs(df).max() / scale < 1e-6, (case, np.abs(df).max() / scale)      # the oracle's point IS the optimum to 1e-6
    f_opt = ref.log_density_x - df                                                # ... and this is it to second order
    err = np.abs(dens - f_opt).max() / scale
    assert err < 5e-6, (case, err)             # (measured: tree 3.9e-6, heavy tails < 3e-6 -- the device stops on a relative
    assert np.std(dens - f_opt) / np.std(f_opt) < 5e-6      #  decrease of 1e-13 of a loss of ~1e6; the reference AS RUN: 4e-5 ... 1e-4)
    # the shortcuts change the iteration path, not the optimum: the plain fp64 solve of the same problem
    monkeypatch.setenv("MELLON_AMD_SUBSAMPLE", "0")
    monkeypatch.setenv("MELLON_AMD_REBUILD", "0")
    plain = mellon.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
    dens_plain = plain.fit_predict(x)
    assert plain.opt_state.success
    e_path = np.abs(dens - dens_plain).max() / scale
    assert e_path < 5e-6, (case, e_path)
    print(f"{case}: device {est.loss_func.n_eval} evaluations / {st['objective_pass_equivalents']:.1f} pass-equivalents, "
          f"{int(st['precond_rebuilds'])} rebuilds, {err:.1e} from the oracle's optimum ({ref.n_eval} oracle evaluations from "
          f"the device's point); plain path {plain.loss_func.n_eval} evaluations, default vs plain {e_path:.1e}")


def test_function_estimator_releases_the_device_copy_of_x(ctx):
    """Advisor (round 5): FunctionEstimator.prepare_inference reaches the shared 1-NN / k-means steps, which upload host
    cells once (`_x_on_device`); the copy must be gone when prepare_inference returns, and copies / pickles of an estimator
    never carry the handle."""
    import mellon_amd as mellon
    n, d = 140_000, 60                                     # 67 MB: above BaseEstimator.DEVICE_X_MIN_BYTES
    rng = np.random.default_rng(3)
    x = rng.normal(size=(n, d))
    y = np.sin(x[:, 0]) + 0.1 * rng.normal(size=n)
    est = mellon.FunctionEstimator(sigma=0.1, n_landmarks=64)
    est.prepare_inference(x)
    assert "_x_dev" not in est.__dict__
    est2 = copy.deepcopy(est)
    assert est2.n_landmarks == est.n_landmarks and "_x_dev" not in est2.__dict__
    de = mellon.DensityEstimator(n_landmarks=64, d=20)     # (d > 50 is refused by the density estimator: state the dimension)
    de.set_x(x)
    dev = de._x_on_device()                                # a held copy ...
    assert dev is not x and "_x_dev" in de.__dict__
    state = pickle.loads(pickle.dumps(de.__getstate__()))  # ... is not part of the state
    assert "_x_dev" not in state
    assert "_x_dev" not in copy.deepcopy(de).__dict__
    de._release_x_on_device()
    assert "_x_dev" not in de.__dict__
    pred = mellon.FunctionEstimator(sigma=0.1, n_landmarks=64).fit_predict(x, y, x[:100])
    assert pred.shape == (100,) and np.all(np.isfinite(pred))


def test_capped_stop_is_reported(ctx, monkeypatch):
    """Advisor (round 5): a solve that ends on its iteration limit while cells are still above the likelihood cap returns
    the capped minorant's loss; mln_map_solve flags it (status bit 2) and the binding strips the bit and logs it."""
    from mellon_amd import cov
    n, d, m = 6000, 10, 200
    x = _tree(n, d, 5)
    nn = mo.exact_nn_distances(x)
    ref_ls, ref_mu = mo.compute_ls(nn), mo.compute_mu(nn, d)
    lm = x[np.sort(np.random.default_rng(7).choice(n, m, replace=False))]
    fit = ctx.fit_prepare(cov.Matern52(ref_ls).lower(d), x, lm, 1e-6)
    V, Vdr = mo.nn_likelihood_constants(nn, d)
    fit.set_likelihood(V, Vdr, ref_mu)
    fit.precond_build()
    z0 = np.full(m, 3.0)                                   # f = L z0 + mu far above: e^{f + V} overshoots the cap everywhere
    z, loss, n_eval, n_iter, status = fit.map_solve(z0, maxiter=1)
    assert (status & 3) in (0, 1, 2)
    loss_true = fit.objective(z)[0]                        # inference.py's objective at the returned point, uncapped
    if status & 4:                                         # flagged: the reported loss is the capped minorant's
        assert (status & 3) != 0 and loss <= loss_true * (1 + 1e-12)
    else:                                                  # not flagged: the reported loss IS the objective's
        assert abs(loss - loss_true) <= 1e-9 * abs(loss_true)
    z2, loss2, _, _, status2 = fit.map_solve(z0)           # and a converged solve never ends capped
    assert status2 == 0
    loss_true2 = fit.objective(z2)[0]
    assert abs(loss2 - loss_true2) <= 1e-9 * abs(loss_true2)
    fit.close()


def test_release_cached_memory_returns_the_pinned_pool(ctx):
    """Advisor (round 5): mln_release_cached_memory() frees the page-locked host blocks too (header: 'all cached blocks')."""
    from mellon_amd import _lib
    est_x = mo.gaussian_mixture(3000, 5, seed=1)
    import mellon_amd as mellon
    mellon.DensityEstimator(n_landmarks=100).fit_predict(est_x)     # a fit pins its m-vector mirrors and returns them to the pool
    _lib.release_cached_memory()
    mellon.DensityEstimator(n_landmarks=100).fit_predict(est_x)     # and the pool fills again from the driver


@pytest.mark.parametrize("d", [10, 50])
def test_pruned_nearest_neighbour_search_is_exact(ctx, d, monkeypatch):
    """The cluster-pruned 1-NN search (2^18 cells and more: coarse clusters, cells sorted by cluster, candidate blocks the
    triangle inequality cannot exclude) returns the exact distances: against a k-d tree (d = 10) and against the unpruned
    device search (both d), with exact duplicates, a tight cluster far from the origin and an outlier among the cells."""
    rng = np.random.default_rng(31)
    n = 300_000 if d == 10 else 1_000_000          # (d = 50 at BASELINE config 3's size: the list re-search's share dealing
    x = mo.gaussian_mixture(n, d, seed=12)         #  once lost ~10 rows in 1e6 to their runner-up, none in 3e5)
    x[1000:1300] = x[5000:5300]                                    # exact duplicates
    x[20000:21000] = 40.0 + 1e-3 * rng.normal(size=(1000, d))      # a tight cluster far away
    x[77] = -500.0                                                 # an outlier: its neighbour is far
    x = np.ascontiguousarray(x)
    pruned = ctx.nn_distances(x)
    monkeypatch.setenv("MELLON_AMD_NN_PRUNE", "0")
    plain = ctx.nn_distances(x)
    monkeypatch.delenv("MELLON_AMD_NN_PRUNE")
    assert np.all(pruned[1000:1300] == 0.0) and np.all(pruned[5000:5300] == 0.0)
    assert np.array_equal(pruned, plain), int(np.sum(pruned != plain))      # the same sums in the same order: the same bits
    assert np.array_equal(ctx.nn_distances(x), pruned)                     # and from call to call
    if d == 10:
        from scipy.spatial import cKDTree
        want = cKDTree(x).query(x, k=2, workers=-1)[0][:, 1]
        assert np.abs(pruned - want).max() <= 1e-9 * want.max()


@pytest.mark.parametrize("case", ["mixture", "tree"])
def test_kmeans_group_bounds(ctx, case, monkeypatch):
    """Lloyd's sweeps with per-stage (group) lower bounds -- 1024 centres and more: the centres swept in a geometric order, the
    cells stored by first label, every open cell against the stages its bounds cannot exclude -- are Lloyd's sweeps: against
    the Hamerly-bound sweeps of the same call (MELLON_AMD_KM_PRUNE=0) the clustering quality is the same (the trajectories
    part where a cell sits within the fp16 pre-filter's error of two centres), the returned centres are a fixed point of
    an EXACT Lloyd step to the stopping tolerance (every cell is with its nearest centre), and the result is the same from
    run to run."""
    n, d, m = 200_000, 12, 1500
    x = mo.gaussian_mixture(n, d, seed=5) if case == "mixture" else _tree(n, d, 9)
    monkeypatch.setenv("MELLON_AMD_KM_LEVELS", "1")
    cg, itg, ing = ctx.kmeans(x, m, seed=11, return_info=True)
    cg2, itg2, ing2 = ctx.kmeans(x, m, seed=11, return_info=True)
    assert itg == itg2 and ing == ing2 and np.array_equal(cg, cg2)
    monkeypatch.setenv("MELLON_AMD_KM_PRUNE", "0")
    cp, itp, inp = ctx.kmeans(x, m, seed=11, return_info=True)
    monkeypatch.delenv("MELLON_AMD_KM_PRUNE")
    assert abs(ing / inp - 1) < 2e-3, (ing, inp)
    assert 0.5 * itp <= itg <= 1.6 * itp, (itg, itp)
    lab = np.concatenate([np.argmin(mo.distance(x[i:i + 20_000], cg), axis=1) for i in range(0, n, 20_000)])
    cnt = np.bincount(lab, minlength=m)
    sums = np.zeros((m, d))
    np.add.at(sums, lab, x)
    means = np.where(cnt[:, None] > 0, sums / np.maximum(cnt, 1)[:, None], cg)
    assert np.sum((means - cg) ** 2) <= 4 * 1e-4 * x.var(axis=0).mean()
    if itg < 300:                                  # converged: inertia of the exact assignment == the reported one
        dmin = np.concatenate([mo.distance(x[i:i + 20_000], cg).min(axis=1) for i in range(0, n, 20_000)])
        assert abs(np.sum(dmin ** 2) / ing - 1) < 1e-9


@pytest.mark.parametrize("n,d,m", [(66_000, 61, 1024), (70_000, 3, 2049), (131_072, 8, 8192)])
def test_kmeans_group_bounds_edges(ctx, n, d, m, monkeypatch):
    """The group-bound sweeps at the edges of where they apply: the widest rows the folded product takes (d = 61), a ragged last
    stage (m = 2049: one centre in the ninth stage), the most stages (m = 8192: 32), n just above the threshold, duplicated
    cells and a far cluster -- each against the Hamerly-bound sweeps of the same call and against an exact Lloyd step."""
    rng = np.random.default_rng(n + d)
    x = mo.gaussian_mixture(n, d, seed=d + 1)
    x[1000:1400] = x[5000:5400]                                    # exact duplicates
    x[20_000:21_000] = 25.0 + 1e-2 * rng.normal(size=(1000, d))    # a tight cluster far away
    x = np.ascontiguousarray(x)
    monkeypatch.setenv("MELLON_AMD_KM_LEVELS", "1")
    cg, itg, ing = ctx.kmeans(x, m, seed=3, max_iter=60, return_info=True)
    monkeypatch.setenv("MELLON_AMD_KM_PRUNE", "0")
    cp, itp, inp = ctx.kmeans(x, m, seed=3, max_iter=60, return_info=True)
    monkeypatch.delenv("MELLON_AMD_KM_PRUNE")
    assert np.all(np.isfinite(cg)) and abs(ing / inp - 1) < 5e-3, (ing, inp)
    # the reported inertia is that of the exact assignment to the returned centres
    dmin = np.concatenate([mo.distance(x[i:i + 8192], cg).min(axis=1) for i in range(0, n, 8192)])
    assert abs(np.sum(dmin ** 2 - 1e-12) / ing - 1) < 1e-8
    # and one exact Lloyd step moves the centres no further than the last sweep did (the sweeps assign every cell to its
    # nearest centre: after 60 of them the movement per sweep is small and shrinking)
    lab = np.concatenate([np.argmin(mo.distance(x[i:i + 8192], cg), axis=1) for i in range(0, n, 8192)])
    cnt = np.bincount(lab, minlength=m)
    sums = np.zeros((m, d))
    np.add.at(sums, lab, x)
    means = np.where(cnt[:, None] > 0, sums / np.maximum(cnt, 1)[:, None], cg)
    lab_p = np.concatenate([np.argmin(mo.distance(x[i:i + 8192], cp), axis=1) for i in range(0, n, 8192)])
    cnt_p = np.bincount(lab_p, minlength=m)
    sums_p = np.zeros((m, d))
    np.add.at(sums_p, lab_p, x)
    means_p = np.where(cnt_p[:, None] > 0, sums_p / np.maximum(cnt_p, 1)[:, None], cp)
    assert np.sum((means - cg) ** 2) <= 3.0 * np.sum((means_p - cp) ** 2) + 1e-12 * x.var(axis=0).sum()
