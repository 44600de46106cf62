"""Round-4 additions on a real MI355X (-m gpu).

* Inputs the solver's shortcuts were NOT tuned on, at the scale where they used to fail (tools/robustness_sweep_large.py):
  1e6 cells of a tree-shaped manifold in 20 nominal dimensions (diffusion-map-like: nearest-neighbour distances over four
  decades -- the Ridge start overflowed, a stopping rule declared convergence at a loss of 1e260, the rebuilt
  preconditioner stalled the solve until the iteration limit) and heavy tails (the rebuild lost positive definiteness).
  No oracle finishes at this size: the default path must agree with the plain path (no subsample phase, no rebuild) of the
  same strictly convex problem and report convergence.
* A start whose loss is not finite (any size): the solve still reaches the oracle's optimum.
* The communicator's accounting (mln_comm_info) on thread-ranks.
"""
import os

import numpy as np
import pytest

from oracle import mellon_oracle as mo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mellon():
    import mellon_amd
    return mellon_amd


@pytest.fixture(scope="module")
def ctx():
    from mellon_amd import _lib
    return _lib.default_context()


def tree_cells(n, d, rng, branches=6):
    """Cells along a branching tree of smooth curves in a 3-D latent space, embedded smoothly in d dimensions whose
    scales decay like a diffusion map's eigenvalues (0.8^k), unevenly populated along pseudo-time."""
    t = rng.beta(0.7, 1.3, size=n)
    b = rng.integers(0, branches, size=n)
    dirs = rng.normal(size=(branches, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    bend = rng.normal(size=(branches, 3)) * 0.5
    z = t[:, None] * dirs[b] + (t ** 2)[:, None] * bend[b] + 0.02 * (1 + 3 * t)[:, None] * rng.normal(size=(n, 3))
    W1 = rng.normal(size=(3, d)); W2 = rng.normal(size=(3, d))
    x = np.tanh(z @ W1) + 0.3 * np.sin(2.0 * z @ W2)
    return np.ascontiguousarray(x * (0.8 ** np.arange(d))[None, :])


def _fit(mellon, xd, lm, nn, monkeypatch, **env):
    for k in ("MELLON_AMD_MIXED", "MELLON_AMD_SUBSAMPLE", "MELLON_AMD_REBUILD", "MELLON_AMD_REBUILD_RANGE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    est = mellon.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
    dens = est.fit_predict(xd)
    st = est._fit.stage_times()
    z = np.asarray(est.pre_transformation)
    loss, grad = est._fit.objective(z)
    est._fit.close()
    return dens, est.opt_state, st, loss, np.abs(grad).max()


@pytest.mark.parametrize("case", ["tree", "heavy tails"])
def test_inputs_the_shortcuts_were_not_tuned_on(mellon, ctx, monkeypatch, case):
    n, d, m = 1_000_000, 20, 2000
    rng = np.random.default_rng(11)
    x = tree_cells(n, d, rng) if case == "tree" else rng.standard_t(3, size=(n, d))
    xd = ctx.to_device(np.ascontiguousarray(x))
    nn = ctx.nn_distances(xd, xd)
    # Landmarks: a seeded subset of the cells.  (Until round 4c: mln_kmeans -- whose fp64 atomics make the centres reproducible
    # to rounding, not bitwise, csrc/kmeans.hip -- so that the guard-off variants below, 500-1000-pass solves that are chaotic
    # in the low-order bits, passed or failed from run to run of the very same code: tools/r04c_tree_inputs_determinism.py.
    # The fit itself is bit-reproducible on identical inputs.)
    lm = np.ascontiguousarray(x[np.sort(np.random.default_rng(42).choice(100_000, m, replace=False))])
    plain = _fit(mellon, xd, lm, nn, monkeypatch, MELLON_AMD_MIXED="0", MELLON_AMD_SUBSAMPLE="0", MELLON_AMD_REBUILD="0")
    assert plain[1].success and np.isfinite(plain[3])
    scale = np.abs(plain[0]).max()
    # (last two: the range guard of the rebuild switched off -- on the tree the rebuilt preconditioner is then garbage, fails
    #  its trial and is replaced by the first one again; in the mixed solve the anchor taken at that point is dropped too)
    for name, env in (("default fp64", {"MELLON_AMD_MIXED": "0"}), ("mixed", {"MELLON_AMD_MIXED": "1"}),
                      ("fp64, rebuild never declines", {"MELLON_AMD_MIXED": "0", "MELLON_AMD_REBUILD_RANGE": "1e300"}),
                      ("mixed, rebuild never declines", {"MELLON_AMD_MIXED": "1", "MELLON_AMD_REBUILD_RANGE": "1e300"})):
        if case != "tree" and "never declines" in name:
            continue                     # (heavy tails: without the guard the whitening loses positive definiteness -- the fallback of its own test)
        dens, state, st, loss, gmax = _fit(mellon, xd, lm, nn, monkeypatch, **env)
        if name == "mixed, rebuild never declines" and not state.success:
            # Guard OFF + 32-bit surrogate: after the garbage preconditioner has failed its trial the solve goes on from where
            # that one left it, on the plain surrogate -- and whether it gets back from there is decided by the low-order bits of
            # the m x m products (round 4c: passes / runs to the iteration limit with either GEMM tiling, run to run of the
            # suite's order: profiles/r04c_tree_guard_off.txt).  What the product owes the caller in that case is the truth:
            assert state.status == 1 and state.nit >= 5000, (case, name, state)
            continue
        assert state.success, (case, name, state)
        assert np.isfinite(dens).all()
        assert abs(loss - plain[3]) <= 1e-9 * abs(plain[3]), (case, name, loss, plain[3])
        # (guard off: a 500-1000-pass solve through a garbage preconditioner and back ends elsewhere in the flat valley of this
        #  problem than the plain solve does -- same loss to 1e-9 above, log-density 1e-5 .. 2.5e-5 apart from run to run,
        #  DESIGN.md S4; the product's paths keep the 1e-5 of north_star)
        assert np.abs(dens - plain[0]).max() <= (5e-5 if "never declines" in name else 1e-5) * scale, (case, name)
    xd.free()


def test_a_start_whose_loss_is_not_finite(mellon):
    """initial_value far outside: e^{f+V} overflows at the first evaluation.  The solver halves the start until the loss is
    finite and below 1e30, then converges to the (unique) optimum."""
    n, d, m = 6000, 6, 150
    x = mo.gaussian_mixture(n, d, seed=41)
    nn = mo.exact_nn_distances(x)
    ref = mo.density_fit(x, n_landmarks=m, nn_distances=nn, lbfgsb_options=mo.LBFGSB_TIGHT)
    for scale in (40.0, 4000.0):
        est = mellon.DensityEstimator(landmarks=ref.landmarks, nn_distances=nn, initial_value=scale * np.ones(m))
        dens = est.fit_predict(x)
        assert est.opt_state.success
        # (rounds 3-4 asserted start_halvings >= 1 here; since round 5 the quadratically continued likelihood keeps the loss
        #  of such a start finite and below 1e30, and the solve walks down from it without halving)
        assert np.abs(dens - ref.log_density_x).max() < 1e-5 * np.abs(ref.log_density_x).max()


def test_comm_info_on_thread_ranks():
    from mellon_amd import distributed

    def body(comm):
        ctx = comm.ctx
        ctx.comm_info(timing=True, reset=True)
        out = ctx.allreduce_sum(np.full(5001, float(comm.rank + 1)))
        big = ctx.allreduce_sum(np.ones(20000))
        info = ctx.comm_info(timing=False)
        return out[0], big[0], info

    res = distributed.run_loopback(3, body)
    for r, (a, b, info) in enumerate(res):
        assert a == 6.0 and b == 3.0
        assert info["transport"] == "loopback" and info["ranks_reported_by_transport"] == 3
        assert info["rank_reported_by_transport"] == r
        assert info["allreduce_calls"] == 2 and info["small_allreduce_calls"] == 1
        assert info["small_allreduce_ms"] > 0.0 and info["large_allreduce_ms"] > 0.0


@pytest.mark.parametrize("knobs", [{"MELLON_AMD_REBUILD_RANGE": "0"},            # every rebuild declines
                                   {"MELLON_AMD_REVERT_AFTER": "1", "MELLON_AMD_MAX_REBUILDS": "1"},   # the one rebuilt preconditioner fails its trial
                                   {"MELLON_AMD_REBUILD_RANGE": "0", "MELLON_AMD_MIXED_MIN_ELEMS": "0", "mixed": "1"}])
def test_rebuild_fallbacks_are_collective(mellon, monkeypatch, knobs):
    """The decline / revert branches issue collectives (or skip them): every rank must take the same one.  Forced here on
    3 thread-ranks at a small size, where the guards would not fire by themselves; the result is the unsharded optimum."""
    from mellon_amd import distributed
    from sklearn.cluster import k_means
    n, d, m = 40_000, 10, 300
    x = mo.gaussian_mixture(n, d, seed=13)
    nn = mo.exact_nn_distances(x)
    lm = np.ascontiguousarray(k_means(x[:8000], m, n_init=1, random_state=42)[0])
    mixed = knobs.get("mixed") == "1"
    monkeypatch.setenv("MELLON_AMD_REBUILD", "1")
    monkeypatch.setenv("MELLON_AMD_MIXED", "1" if mixed else "0")
    for k, v in knobs.items():
        if k != "mixed":
            monkeypatch.setenv(k, v)
    est = mellon.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
    single = est.fit_predict(x)
    st = est._fit.stage_times()
    if "MELLON_AMD_REVERT_AFTER" in knobs:
        assert int(st["precond_reverts"]) == 1
    else:
        # (round 5: a solve may ask for a rebuild more than once; after two attempts that came to nothing it stops asking)
        assert 1 <= int(st["precond_rebuilds_declined"]) <= 2 and int(st["precond_rebuilds"]) == 0
    assert est.opt_state.success

    def body(comm):
        lo, hi = distributed.shard_bounds(n, comm.world_size, comm.rank)
        e = mellon.DensityEstimator(landmarks=lm, nn_distances=nn[lo:hi], check_rank=False)
        dens = e.fit_predict(np.ascontiguousarray(x[lo:hi]))
        s = e._fit.stage_times()
        return dens, e.opt_state.success, int(s["precond_reverts"]), int(s["precond_rebuilds_declined"])

    parts = distributed.run_loopback(3, body)
    assert all(p[1] for p in parts)
    assert len({(p[2], p[3]) for p in parts}) == 1
    sharded = np.concatenate([p[0] for p in parts])
    assert np.abs(sharded - single).max() < 2e-6 * np.abs(single).max()


def test_kmeans_coarse_to_fine(ctx, monkeypatch):
    """Above 64 cells per centre mln_kmeans seeds and pre-converges on every s-th cell and only polishes on all of them:
    same quality (inertia) as the one-level run, every centre the mean of its members, reproducible for a fixed seed."""
    x = mo.gaussian_mixture(260_000, 10, seed=3)
    m = 500
    c2, it2, inertia2 = ctx.kmeans(x, m, seed=42, return_info=True)
    monkeypatch.setenv("MELLON_AMD_KM_LEVELS", "1")
    c1, it1, inertia1 = ctx.kmeans(x, m, seed=42, return_info=True)
    monkeypatch.delenv("MELLON_AMD_KM_LEVELS")
    assert inertia2 < 1.01 * inertia1, (inertia2, inertia1)
    lab = np.argmin(mo.distance(x[:20000], c2), axis=1)
    assert np.isfinite(c2).all() and len(np.unique(lab)) > 0.9 * m
    c2b = ctx.kmeans(x, m, seed=42)
    assert np.abs(c2b - c2).max() < 1e-9 * np.abs(c2).max()


def test_first_fit_of_a_process_equals_the_later_ones(mellon):
    """Whether the second preconditioner is built is decided from a cost MODEL of the build (csrc/api_solve.hip), not from the
    stopwatch on the first one: a cold process (slow first build -> no rebuild -> another iteration path -> other last digits at
    the default stopping rule) and a warm one take the same decisions.  Two fits of the same data are the same bits, with the
    same evaluation count -- also when the library has just been handed a huge unrelated problem (allocator state)."""
    rng = np.random.default_rng(0)
    x = rng.normal(size=(300, 3)) @ np.array([[2.0, 0.0, 0.0], [1.0, 1.0, 0.0], [0.3, 0.2, 0.5]]).T
    outs, evals = [], []
    for rep in range(3):
        est = mellon.DensityEstimator(n_landmarks=40)
        outs.append(est.fit_predict(x))
        evals.append(est.loss_func.n_eval)
        est._fit.close()
        if rep == 0:
            big = mo.gaussian_mixture(30000, 10, seed=1)
            mellon.DensityEstimator(n_landmarks=300).fit_predict(big)
    assert evals[0] == evals[1] == evals[2], evals
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2])


def test_kmeans_distance_bounds(ctx, monkeypatch):
    """Lloyd's sweeps with Hamerly's distance bounds skip the cells whose assignment cannot have changed: the same
    algorithm, so the sweeps, the inertia and the centres are those of the plain sweeps (up to cells that sit within the
    fp16 pre-filter's error of two centres, which the plain sweep may give to either), and every centre is the mean of its
    cells although the sums only follow the cells that move."""
    x = mo.gaussian_mixture(260_000, 12, seed=5)
    m = 400
    monkeypatch.setenv("MELLON_AMD_KM_LEVELS", "1")          # one level: the bounds are the only difference
    cb, itb, inb = ctx.kmeans(x, m, seed=11, return_info=True)
    monkeypatch.setenv("MELLON_AMD_KM_BOUNDS", "0")
    cp, itp, inp = ctx.kmeans(x, m, seed=11, return_info=True)
    monkeypatch.delenv("MELLON_AMD_KM_BOUNDS")
    assert abs(inb / inp - 1) < 2e-3, (inb, inp)
    # (the sweep COUNT to sklearn's tolerance is a property of the trajectory, not of the algorithm: cells within the pre-filter's
    #  error of two centres may go to either, and from there the counts drift -- measured over seeds 11 / 12 / 13 with bounds
    #  against without: 209 / 259 / 300 against 209 / 276 / 229 with the round-5 sweep kernel, 261 / 283 / 272 with round 6's)
    assert 0.5 * itp <= itb <= 1.6 * itp, (itb, itp)
    lab = np.argmin(mo.distance(x, cb), axis=1)
    means = np.stack([x[lab == j].mean(axis=0) if np.any(lab == j) else cb[j] for j in range(m)])
    # one more Lloyd step from the returned centres moves them by no more than the stopping tolerance allows
    assert np.sum((means - cb) ** 2) <= 4 * 1e-4 * x.var(axis=0).mean()
    # two levels (the default at this size) with bounds on both
    monkeypatch.delenv("MELLON_AMD_KM_LEVELS")
    c2, it2, in2 = ctx.kmeans(x, m, seed=11, return_info=True)
    assert in2 < 1.01 * inp


def test_nn_open_rows_resolved_from_a_candidate_list(ctx, monkeypatch):
    """Rows the fp64 certification leaves open (runner-up within the pre-filter's error of the winner) are resolved by a
    second fp16 sweep that lists every candidate below the row's threshold and the exact distances of the listed pairs --
    the distances of the exact fp64 re-search it replaces, and of a tree search; a list that outgrows its buffer
    (a tight cluster of thousands of mutual near-ties) falls back to that re-search."""
    rng = np.random.default_rng(21)
    n, d = 60000, 50
    x = mo.gaussian_mixture(n, d, seed=9)
    x[1000:1400] = x[2000:2400] + 3e-4 * rng.normal(size=(400, d))      # near-ties: many open rows, a few candidates each
    x[5000:5010] = x[5010:5020]                                          # exact duplicates
    monkeypatch.setenv("MELLON_AMD_NN_PREFILTER_MIN", "1")
    listed = ctx.nn_distances(x)
    monkeypatch.setenv("MELLON_AMD_NN_LIST", "0")
    exact = ctx.nn_distances(x)
    monkeypatch.delenv("MELLON_AMD_NN_LIST")
    want = mo.exact_nn_distances(x)
    # (the exact re-search picks its winner from |x|^2 - 2 x.y + |y|^2 and reports that candidate's direct distance; the list
    #  takes the minimum of the direct distances: candidates closer together than fp64 cancellation may swap)
    nz = want > 0
    assert np.abs(listed[nz] / exact[nz] - 1).max() < 1e-12 and np.array_equal(listed == 0, exact == 0)
    assert np.abs(listed / np.maximum(want, 1e-300) - 1)[want > 0].max() < 1e-12 and np.all(listed[5000:5020] == 0)
    # overflow of the list: 4000 cells within the error bound of each other
    y = x.copy()
    y[:4000] = 30.0 + 1e-4 * rng.normal(size=(4000, d))
    got = ctx.nn_distances(y)
    wanty = mo.exact_nn_distances(y)
    tol2 = 64 * np.finfo(float).eps * np.linalg.norm(y, axis=1).max() ** 2
    assert np.abs(got**2 - wanty**2).max() < tol2
    # ... and so many that the pair counter would pass 2^31 (50 000 mutual near-ties: 2.5e9 pairs; this was a write fault
    # through a wrapped 32-bit slot index until the sweep stopped feeding an abandoned list)
    z = np.concatenate([4.0 + 0.01 * rng.normal(size=(50_000, 20)), mo.gaussian_mixture(30_000, 20, seed=4)])
    gz = ctx.nn_distances(z)
    wz = mo.exact_nn_distances(z)
    assert np.all(np.isfinite(gz)) and np.abs(gz**2 - wz**2).max() < 64 * np.finfo(float).eps * np.linalg.norm(z, axis=1).max() ** 2


def test_tree_data_sharded_at_scale(mellon, ctx, monkeypatch):
    """The guards of the rebuild (decline / trial / revert) decide from all-reduced numbers and the replicated solver state:
    1e6 tree-shaped cells on 2 thread-ranks reach the unsharded fit's optimum and report the same decisions."""
    from mellon_amd import distributed
    n, d, m = 1_000_000, 20, 2000
    rng = np.random.default_rng(11)
    x = tree_cells(n, d, rng)
    xd = ctx.to_device(x)
    nn = ctx.nn_distances(xd, xd)
    xd.free()
    lm = ctx.kmeans(x[:100_000], m, seed=42)
    monkeypatch.setenv("MELLON_AMD_MIXED", "0")
    est = mellon.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
    single = est.fit_predict(x)
    st1 = est._fit.stage_times()
    assert est.opt_state.success
    est._fit.close()

    def body(comm):
        lo, hi = distributed.shard_bounds(n, comm.world_size, comm.rank)
        e = mellon.DensityEstimator(landmarks=lm, nn_distances=nn[lo:hi], check_rank=False)
        dens = e.fit_predict(np.ascontiguousarray(x[lo:hi]))
        s = e._fit.stage_times()
        out = dens, e.opt_state.success, (int(s["precond_rebuilds"]), int(s["precond_rebuilds_declined"]), int(s["precond_reverts"]))
        e._fit.close()
        return out

    parts = distributed.run_loopback(2, body)
    assert all(p[1] for p in parts)
    assert parts[0][2] == parts[1][2]
    sharded = np.concatenate([p[0] for p in parts])
    assert np.abs(sharded - single).max() < 1e-5 * np.abs(single).max()


# ---- round 4c: the GEMM's mixed-tile pipelined kernel -------------------------------------------------------------------
@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
def test_gemm_mixed_tiles_same_bits_as_single_size(ta, tb):
    """k_dgemm_mix (whole rounds of 128-tiles, the stragglers as 64-wide quadrants, two LDS buffers) sums every element's
    k in the order the single-size kernels do: the results are the same bits, for every tile set and K-range mode the
    factorisations and the preconditioner build use (decomposition.py:111-123, conditional.py:57-66)."""
    from mellon_amd import _lib
    ctx = _lib.default_context()
    cases = [  # M, N, K, lower_only, kmode, beta
        (700, 900, 300, 0, 0, 0.0),      # ragged edges in both directions, K not a multiple of 16
        (645, 645, 517, 1, 0, -1.0),     # lower triangle of tiles (Cholesky trailing update), beta in the accumulators
        (645, 645, 645, 2, 0, 0.0),      # strictly lower
        (645, 645, 645, 3, 0, 0.0),      # upper
        (768, 768, 768, 0, 1, 0.0),      # block-diagonal op(A)
        (768, 700, 768, 0, 2, 1.0),      # block-diagonal op(B)
        (900, 900, 900, 0, 3, 0.0),      # k <= row
        (900, 900, 900, 0, 4, 0.0),      # k <= column
        (900, 900, 900, 1, 7, 0.0),      # column <= k <= row, lower tiles
        (130, 70, 40, 0, 0, 0.5),        # two tiles
        (645, 645, 512, 1, 0, -1.0),     # whole k-tiles: the quadrants keep four of them in flight (gemm_tile64_ring), odd edges
        (650, 645, 256, 0, 0, 0.0),      # the same with an odd column count in the operand that is staged along it
        (700, 389, 304, 0, 0, 2.0),
        (1280, 1280, 1280, 1, 7, 0.0),   # K ranges of 128 .. 1280 under the ring
        (64, 64, 8, 0, 0, 0.0),          # one quadrant, one partial k-tile
    ]
    for (M, N, K, lo, km, beta) in cases:
        if km in (1, 3, 7) and K != M:
            continue
        d, v = ctx.diag_dgemm_compare(ta, tb, M, N, K, lower_only=lo, kmode=km, beta=beta, any_size=True)
        assert v > 0.0 and d == 0.0, (M, N, K, lo, km, beta, d, v)
    # at the size where the policy itself selects the mixed kernel (more than one round of 128-tiles on this GPU)
    d, v = ctx.diag_dgemm_compare(ta, tb, 3200, 3100, 200, lower_only=0, kmode=0, beta=0.0, any_size=False)
    assert v > 0.0 and d == 0.0
    d, v = ctx.diag_dgemm_compare(ta, tb, 4500, 4500, 256, lower_only=1, kmode=0, beta=-1.0, any_size=False)
    assert v > 0.0 and d == 0.0


def test_kmeans_landmarks_are_the_same_bits_every_time(ctx):
    """mln_kmeans summed its clusters with fp64 atomics until round 4c: centres reproducible to rounding only, and with them the
    landmarks of every default call on a large input (parameters.py:243-291 hands the reference's k-means a fixed
    random_state for exactly this reason).  The sums are fixed-point integers now (csrc/kmeans.hip): any order, the same bits."""
    rng = np.random.default_rng(5)
    x = np.ascontiguousarray(rng.normal(size=(120_000, 12)) * (0.7 ** np.arange(12))[None, :] + rng.integers(0, 7, size=(120_000, 1)))
    a = ctx.kmeans(x, 700, seed=3)
    b = ctx.kmeans(x, 700, seed=3)
    assert a.shape == (700, 12) and np.isfinite(a).all()
    assert np.array_equal(a, b)
    # the fixed point loses nothing that matters: every centre is the mean of its cells to 1e-12 of the column's range
    d2 = (x * x).sum(1)[:, None] - 2.0 * x @ a.T + (a * a).sum(1)[None, :]
    lab = d2.argmin(1)
    worst = 0.0
    for j in np.unique(lab)[:50]:
        worst = max(worst, np.abs(x[lab == j].mean(0) - a[j]).max())
    assert worst < 1e-3 * np.abs(x).max()          # (Lloyd's stopping rule leaves the centres a sweep away from their means)
