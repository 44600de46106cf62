"""world_size-2 `gloo` tests (CPU) of the cell-sharded path: shard layout, global heuristics,
and the exchange contract the C-ABI implements on RCCL (sum of per-shard likelihood terms
all-reduced, prior added once; Ridge Gram all-reduced)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gloo_host_comm(dist):
    """The product's host communicator interface on torch.distributed (gloo) -- tests only; the product itself
    exchanges its few host-side kilobytes over a plain socket (distributed.SocketHostComm)."""
    from mellon_amd import distributed

    class GlooHostComm(distributed.HostComm):
        rank, world_size = dist.get_rank(), dist.get_world_size()

        def allgather(self, obj):
            out = [None] * self.world_size
            dist.all_gather_object(out, obj)
            return out

    return GlooHostComm()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import scipy.linalg as sla
    from threadpoolctl import threadpool_limits
    threadpool_limits(1)       # single-threaded BLAS: every rank must hold bit-identical replicated data
    from mellon_amd import distributed, parameters
    from oracle import mellon_oracle as mo

    comm = distributed.set_current(distributed.ShardedCommunicator(None, _gloo_host_comm(dist)))
    assert (comm.rank, comm.world_size) == (rank, world)
    n, d, m = 4001, 6, 60                                   # odd n: uneven shards
    x = mo.gaussian_mixture(n, d, seed=21)                  # identical on every rank
    nn = mo.exact_nn_distances(x)
    lo, hi = distributed.shard_bounds(n, world, rank)
    xs, nns = x[lo:hi], nn[lo:hi]

    # heuristics on shards == heuristics on all cells (parameters.py:599,613)
    assert abs(parameters.compute_mu(nns, d) - mo.compute_mu(nn, d)) < 1e-12
    assert abs(parameters.compute_ls(nns) - mo.compute_ls(nn)) < 1e-12 * mo.compute_ls(nn)
    assert np.array_equal(comm.allgather_rows(nns), nn)
    assert comm.broadcast(b"unique-id" if rank == 0 else None) == b"unique-id"
    assert comm.global_count(hi - lo) == n
    for q in (0.01, 0.5, 0.97, 0.0, 1.0):                   # partial-gather quantile == quantile of all cells
        assert comm.global_quantile(nns, q) == pytest.approx(np.quantile(nn, q), rel=1e-14, abs=0)

    ref = mo.density_fit(x, n_landmarks=m, nn_distances=nn, lbfgsb_options=mo.LBFGSB_TIGHT)
    cov, mu, lm = ref.cov_func, ref.mu, ref.landmarks
    Ls_own = mo.standard_low_rank(xs, cov, lm, Lp=ref.Lp)   # this rank's rows of L from its own cells; Lp replicated
    assert np.abs(Ls_own - ref.L[lo:hi]).max() < 1e-8 * np.abs(ref.L).max()
    Ls = ref.L[lo:hi]      # bit-identical rows for the exchange-contract checks below (recomputed rows differ
                           # by cond(Lp)-amplified rounding, which the optimum check at the end absorbs)
    Vs, Vdrs = mo.nn_likelihood_constants(nns, d)

    # Ridge: all-reduce of the m x m Gram and of L^T t, then the replicated solve
    G = comm.allreduce_sum(Ls.T @ Ls)
    rhs = comm.allreduce_sum(Ls.T @ (mo.mle(nns, d) - mu))
    C = sla.cholesky(G + np.eye(m), lower=True)
    z0 = sla.solve_triangular(C.T, sla.solve_triangular(C, rhs, lower=True), lower=False)
    assert np.abs(z0 - ref.initial_value).max() < 1e-9 * np.abs(ref.initial_value).max()

    # objective: per-shard likelihood terms summed across ranks, prior added once
    def sharded(z):
        f = Ls @ z + mu
        a = np.exp(f + Vs)
        part = np.concatenate([[-np.sum((f + Vdrs) - a)], Ls.T @ (a - 1.0)])
        tot = comm.allreduce_sum(part)
        return tot[0] + 0.5 * z @ z + 0.5 * m * np.log(2 * np.pi), tot[1:] + z

    V, Vdr = mo.nn_likelihood_constants(nn, d)
    for z in (ref.initial_value, ref.pre_transformation):
        l_s, g_s = sharded(z)
        l_g, g_g = mo.loss_and_grad(z, ref.L, mu, V, Vdr)
        assert abs(l_s - l_g) < 1e-9 * abs(l_g), (l_s, l_g)      # re-association of sums whose terms reach 1e3
        assert np.abs(g_s - g_g).max() < 1e-7 * max(np.abs(g_g).max(), 1.0), np.abs(g_s - g_g).max()

    # sparse Nystroem (decomposition.py:213-266) shards the same way: the m x m Gram B^T B is all-reduced, its
    # eigenpairs are replicated, and every rank projects its own rows  L_rows = B_rows U[:, -p:]
    from mellon_amd.decomposition import _select_rank
    Sg, Ug = np.linalg.eigh(comm.allreduce_sum(Ls.T @ Ls))
    p = _select_rank(Sg, 0.99)
    ref_nys = mo.modified_low_rank(x, cov, lm, rank=0.99)
    assert ref_nys.shape[1] == p
    rows = Ls @ Ug[:, -p:]
    assert np.abs(rows @ rows.T - ref_nys[lo:hi] @ ref_nys[lo:hi].T).max() < 1e-9 * np.abs(ref_nys @ ref_nys.T).max()

    # landmark conditional of the function estimator (config 5; conditional.py:57-66,526-545,660-685): A A^T and
    # A (y - mu) are all-reduced once; with one sigma per output the replicated eigendecomposition of A A^T serves
    # every level, and the leverage of a rank's own cells comes from the all-reduced  L^T L + jitter Lp^-1 Lp^-T
    pq = 5
    rng = np.random.default_rng(3)
    Yall = np.sin(x @ rng.normal(size=(d, pq))) + 0.1 * rng.normal(size=(n, pq))
    sig = np.array([0.3, 0.5, 0.3, 1.0, 2.0])
    want = mo.landmarks_conditional(x, lm, Yall, 0.0, cov, Lp=ref.Lp, sigma=sig, with_uncertainty=True)   # carries L = Lp
    G0 = comm.allreduce_sum(Ls.T @ Ls)
    C0 = comm.allreduce_sum(Ls.T @ Yall[lo:hi])
    lam, U = np.linalg.eigh(G0)
    Wsh = sla.solve_triangular(ref.Lp.T, U @ ((U.T @ (C0 / sig ** 2)) / (np.maximum(lam, 0)[:, None] / sig ** 2 + 1)),
                               lower=False)
    assert np.abs(Wsh - want.weights).max() < 1e-8 * np.abs(want.weights).max()
    Li = sla.solve_triangular(ref.Lp, np.eye(m), lower=True)
    theta, Vv = np.linalg.eigh(G0 + 1e-6 * (Li @ Li.T))
    h_own = ((Ls @ Vv) ** 2) @ (1.0 / (sig[None, :] ** 2 + theta[:, None]))
    assert np.abs(h_own - want.leverage(x)[lo:hi]).max() < 1e-7

    Ls = Ls_own                                             # the real thing: every rank factors its own cells
    res = mo.minimize_lbfgsb(sharded, z0, mo.LBFGSB_TIGHT)  # every rank runs the same host optimiser
    dens = comm.allgather_rows(Ls @ res.pre_transformation + mu)
    err = np.abs(dens - ref.log_density_x).max() / np.abs(ref.log_density_x).max()
    assert err < 1e-6, err
    with open(os.path.join(out_dir, f"ok{rank}"), "w") as fh:
        fh.write(repr(err))
    dist.destroy_process_group()


def test_cell_sharded_contract_world_size_2(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def _socket_worker(rank, world, name, q):
    sys.path.insert(0, ROOT)
    from mellon_amd import distributed
    host = distributed.SocketHostComm(("unix", name), rank, world, timeout=60.0)
    comm = distributed.ShardedCommunicator(None, host)
    got = comm.allreduce_sum(np.arange(5.0) * (rank + 1))
    rows = comm.allgather_rows(np.full(rank + 1, float(rank)))
    uid = comm.broadcast(b"x" * 128 if rank == 0 else None)
    comm.barrier()
    quant = comm.global_quantile(np.arange(rank, 1000, world, dtype=np.float64), 0.01)
    host.close()
    q.put((rank, got.tolist(), rows.tolist(), uid, quant))


def test_socket_host_communicator_three_ranks():
    """The product's own host-side exchange (standard library sockets, no framework): three processes."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = f"mellon_amd.test.{os.getpid()}"
    procs = [ctx.Process(target=_socket_worker, args=(r, 3, name, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, red, rows, uid, quant in got:
        assert red == (np.arange(5.0) * 6).tolist()
        assert rows == [0.0, 1.0, 1.0, 2.0, 2.0, 2.0]
        assert uid == b"x" * 128
        assert quant == pytest.approx(np.quantile(np.arange(1000.0), 0.01), rel=1e-14)


def test_thread_host_communicator():
    """Thread-ranks (the host side of the loopback communicator) see the same reductions."""
    import threading
    sys.path.insert(0, ROOT)
    from mellon_amd import distributed
    group = distributed.ThreadGroup(4)
    out = [None] * 4

    def body(r):
        comm = distributed.ShardedCommunicator(None, distributed.ThreadHostComm(group, r))
        distributed.set_thread_current(comm)
        assert distributed.current() is comm
        out[r] = (comm.allreduce_sum(np.array([r + 1.0])), comm.global_quantile(np.arange(r, 400, 4.0), 0.25))
        distributed.set_thread_current(None)

    ts = [threading.Thread(target=body, args=(r,)) for r in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(o[0][0] == 10.0 for o in out)
    assert all(o[1] == pytest.approx(np.quantile(np.arange(400.0), 0.25)) for o in out)
    assert distributed.current().world_size == 1


def test_product_has_no_framework_dependency():
    """north_star: "no PyTorch" -- nothing under mellon_amd/ imports torch (the gloo communicator lives in tests/)."""
    import re
    pkg = os.path.join(ROOT, "mellon_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            assert not re.search(r"^\s*(import|from)\s+torch\b", open(os.path.join(pkg, f)).read(), re.M), f


def test_sharded_estimator_requires_shared_inputs():
    """Landmarks and nn_distances need all cells: in sharded mode they must be passed in."""
    sys.path.insert(0, ROOT)
    import mellon_amd
    from mellon_amd import distributed

    class Two(distributed.Communicator):
        rank, world_size = 0, 2

    distributed.set_current(Two())
    try:
        est = mellon_amd.DensityEstimator(n_landmarks=10)
        est.set_x(np.zeros((40, 2)))
        with pytest.raises(NotImplementedError):
            est._compute_nn_distances()
        with pytest.raises(NotImplementedError):
            est._compute_landmarks()
    finally:
        distributed.set_current(distributed.Communicator())
