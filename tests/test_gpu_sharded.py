"""The cell-sharded path, N > 1, on ONE GPU: N thread-ranks joined by the library's loopback communicator run the
product's real sharded code -- per-rank kernel matrix, all-reduced Ridge Gram from a rank-dependent row sample,
per-evaluation all-reduce of [r ; lik] inside the device-resident solver, replicated m x m factor work, global
heuristics over the host communicator -- and must reproduce the unsharded fit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workload():
    from oracle import mellon_oracle as mo
    from sklearn.cluster import k_means
    n, d, m = 24_000, 10, 300
    x = mo.gaussian_mixture(n, d, seed=11)
    nn = mo.exact_nn_distances(x)
    lm = k_means(x[:6000], m, n_init=1, random_state=42)[0]
    return x, nn, np.ascontiguousarray(lm)


def _fit_single(x, nn, lm, **kw):
    import mellon_amd
    est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, **kw)
    dens = est.fit_predict(x)
    return est, dens


def _fit_sharded(n_ranks, x, nn, lm, **kw):
    import mellon_amd
    from mellon_amd import distributed

    def body(comm):
        lo, hi = distributed.shard_bounds(x.shape[0], comm.world_size, comm.rank)
        est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn[lo:hi], **kw)
        dens = est.fit_predict(np.ascontiguousarray(x[lo:hi]))
        pred = est.predict(x[:64])
        return dict(dens=dens, z=np.array(est.pre_transformation), mu=est.mu, ls=est.ls, n_eval=est.loss_func.n_eval,
                    pred=pred, rank=comm.rank)

    return distributed.run_loopback(n_ranks, body)


def test_loopback_collectives():
    """all-reduce / broadcast of the loopback group: rank-order sums, identical bits on every rank."""
    from mellon_amd import distributed
    rng = np.random.default_rng(0)
    parts = rng.normal(size=(4, 10_001))

    def body(comm):
        out = comm.ctx.allreduce_sum(parts[comm.rank])
        big = comm.ctx.allreduce_sum(np.full(3_000_000, float(comm.rank + 1)))
        return out, float(big[0]), float(big[-1])

    res = distributed.run_loopback(4, body)
    want = ((parts[0] + parts[1]) + parts[2]) + parts[3]
    for out, b0, b1 in res:
        assert np.array_equal(out, want)
        assert (b0, b1) == (10.0, 10.0)


def test_loopback_error_does_not_hang():
    """A rank failing outside a collective releases the ranks waiting inside one."""
    from mellon_amd import distributed

    def body(comm):
        if comm.rank == 1:
            raise RuntimeError("rank 1 gives up")
        return comm.ctx.allreduce_sum(np.ones(8))

    with pytest.raises(RuntimeError, match="gives up"):
        distributed.run_loopback(3, body)


@pytest.mark.parametrize("n_ranks", [2, 4, 8])
def test_sharded_fit_equals_unsharded(workload, n_ranks):
    """N = 2, 4, 8 shards of the same cells through mln_fit_prepare / mln_precond_build / mln_ridge_init /
    mln_map_solve / mln_transform with real collectives: log-density == the single-rank fit."""
    x, nn, lm = workload
    tight = dict(ftol=1e-15, gtol=1e-10)       # both runs at the optimum: what is left is the re-association of sums
    est1, dens1 = _fit_single(x, nn, lm)
    est1t = None
    import mellon_amd
    est1t = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn)
    est1t.lbfgsb_options = tight
    dens1t = est1t.fit_predict(x)
    res = _fit_sharded(n_ranks, x, nn, lm)
    dens = np.concatenate([r["dens"] for r in res])
    scale = np.abs(dens1).max()
    assert all(r["mu"] == est1.mu and r["ls"] == est1.ls for r in res)          # global heuristics, bit for bit
    assert all(np.array_equal(r["z"], res[0]["z"]) for r in res)                 # one optimiser state on every rank
    assert all(np.array_equal(r["pred"], res[0]["pred"]) for r in res)
    assert np.abs(dens - dens1).max() / scale < 1e-6                              # default stopping rule, both sides
    assert np.abs(res[0]["pred"] - dens1[:64]).max() / scale < 1e-6

    def tight_body(comm):
        import mellon_amd
        from mellon_amd import distributed
        lo, hi = distributed.shard_bounds(x.shape[0], comm.world_size, comm.rank)
        est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn[lo:hi])
        est.lbfgsb_options = tight
        return est.fit_predict(np.ascontiguousarray(x[lo:hi]))

    from mellon_amd import distributed
    denst = np.concatenate(distributed.run_loopback(n_ranks, tight_body))
    assert np.abs(denst - dens1t).max() / scale < 2e-7, np.abs(denst - dens1t).max() / scale   # measured 3e-8: the floor set by the re-association of the sums (|loss| ~ 1e5, eps 1e-16)


def test_sharded_fit_mixed_precision_and_uneven_shards(workload, monkeypatch):
    """The fp32 warm-up passes (forced at this size) and shards of unequal length (n not divisible by N)."""
    x, nn, lm = workload
    x, nn = x[:-5], nn[:-5]
    monkeypatch.setenv("MELLON_AMD_MIXED", "1")
    monkeypatch.setenv("MELLON_AMD_MIXED_MIN_ELEMS", "1")
    est1, dens1 = _fit_single(x, nn, lm)
    res = _fit_sharded(3, x, nn, lm)
    stats = est1._fit.stage_times()
    assert stats["objective32_launches"] > 0
    dens = np.concatenate([r["dens"] for r in res])
    assert dens.shape == dens1.shape
    assert np.abs(dens - dens1).max() / np.abs(dens1).max() < 1e-6


def test_sharded_fit_matches_oracle(workload):
    """... and the sharded result is the oracle's optimum (not merely self-consistent)."""
    from oracle import mellon_oracle as mo
    x, nn, lm = workload
    ref = mo.density_fit(x, landmarks=lm, nn_distances=nn, lbfgsb_options=mo.LBFGSB_TIGHT)
    res = _fit_sharded(4, x, nn, lm)
    dens = np.concatenate([r["dens"] for r in res])
    assert abs(res[0]["mu"] - ref.mu) < 1e-12 and abs(res[0]["ls"] - ref.ls) < 1e-12 * ref.ls
    assert np.abs(dens - ref.log_density_x).max() / np.abs(ref.log_density_x).max() < 1e-5
