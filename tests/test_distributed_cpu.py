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
    # the star reduction that carries host-staged device collectives, through the library's callback signature
    import ctypes
    staged = distributed.HostStagedCollectives(host)
    buf = (np.arange(7.0) + 100.0 * rank)
    gathered = np.zeros(7 * world)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p).value
    assert staged(None, 0, ptr(buf), None, 7) == 0            # all-reduce in place
    red = buf.copy()
    buf2 = np.full(3, float(rank))
    assert staged(None, 1, ptr(buf2), None, 3) == 0           # broadcast from rank 0
    buf3 = np.full(7, float(rank))
    assert staged(None, 2, ptr(buf3), ptr(gathered), 7) == 0  # all-gather
    assert staged(None, 9, ptr(buf3), None, 7) == 1 and isinstance(staged.failure, ValueError)
    host.close()
    q.put((rank, got.tolist(), rows.tolist(), uid, quant, red.tolist(), buf2.tolist(), gathered.tolist()))


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
    for rank, red, rows, uid, quant, staged_sum, staged_bc, staged_ag in got:
        assert red == (np.arange(5.0) * 6).tolist()
        assert rows == [0.0, 1.0, 1.0, 2.0, 2.0, 2.0]
        assert uid == b"x" * 128
        assert quant == pytest.approx(np.quantile(np.arange(1000.0), 0.01), rel=1e-14)
        assert staged_sum == (3 * np.arange(7.0) + 300.0).tolist()
        assert staged_bc == [0.0, 0.0, 0.0]
        assert staged_ag == np.repeat(np.arange(3.0), 7).tolist()


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


def test_sharded_estimator_gathers_shared_inputs(monkeypatch):
    """Landmarks and nn_distances need all cells (parameters.py:243-291,352-433): a sharded fit gathers the cells over the
    host communicator -- rank 0 clusters and broadcasts, every rank searches its own cells among all of them.  Two
    thread-ranks over the product's ThreadHostComm; the device search is replaced by a NumPy one (no GPU here)."""
    sys.path.insert(0, ROOT)
    import threading
    import mellon_amd
    from mellon_amd import _lib, distributed, parameters
    from oracle import mellon_oracle as mo

    class FakeCtx:
        n_ranks = 2

        def nn_distances(self, x, y=None, self_offset=0):
            y = x if y is None else y
            d2 = ((x[:, None, :] - y[None, :, :]) ** 2).sum(-1)
            d2[np.arange(x.shape[0]), np.arange(x.shape[0]) + self_offset] = np.inf
            return np.sqrt(d2.min(axis=1))

    monkeypatch.setattr(_lib, "default_context", lambda: FakeCtx())
    n, d, m = 301, 3, 12
    x = mo.gaussian_mixture(n, d, seed=4)
    times = np.repeat(np.arange(3.0), [100, 100, 101])
    xt = np.column_stack([x, times])
    nn_all = mo.exact_nn_distances(x)
    nn_t = mo.per_time_nn_distances(x, times)
    group = distributed.ThreadGroup(2)
    out, errs = [None, None], []

    def body(rank):
        try:
            comm = distributed.ShardedCommunicator(None, distributed.ThreadHostComm(group, rank))
            distributed.set_thread_current(comm)
            lo, hi = distributed.shard_bounds(n, 2, rank)
            est = mellon_amd.DensityEstimator(n_landmarks=m)
            est.set_x(np.ascontiguousarray(x[lo:hi]))
            est.gp_type = mellon_amd.GaussianProcessType.SPARSE_CHOLESKY
            nn = est._compute_nn_distances()
            lm = est._compute_landmarks()
            tse = mellon_amd.TimeSensitiveDensityEstimator(n_landmarks=m, ls_time=1.0)
            tse.set_x(np.ascontiguousarray(xt[lo:hi]))
            tse.d = d
            nn_time = tse._nn_within_time_points(False)
            avg = parameters.compute_average_cell_count(tse.x, False)
            out[rank] = (nn, lm, nn_time, avg)
        except BaseException as e:     # noqa: BLE001
            errs.append(e)
            group.barrier.abort()
        finally:
            distributed.set_thread_current(None)

    ts = [threading.Thread(target=body, args=(r,)) for r in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    assert np.allclose(np.concatenate([out[0][0], out[1][0]]), nn_all, rtol=1e-10)
    assert np.array_equal(out[0][1], out[1][1]) and out[0][1].shape == (m, d)       # the same landmarks, bit for bit
    assert np.allclose(np.concatenate([out[0][2], out[1][2]]), nn_t, rtol=1e-10)     # within time points, across ranks
    assert out[0][3] == out[1][3] == n / 3


# ---- the host communicator's wire format and handshake (no GPU, no framework) ---------------------------------------
def test_host_codec_round_trip_and_refusals():
    from mellon_amd import distributed as D
    rng = np.random.default_rng(0)
    samples = [None, True, False, 0, -7, 2 ** 40, 1.5, float("inf"), "landmarks: k-means, untimed", b"\x00\x01" * 64,
               rng.normal(size=(5, 3)), rng.normal(size=0), np.arange(6, dtype=np.int64).reshape(2, 3),
               np.array([True, False]), rng.normal(size=(2, 3, 4)).astype(np.float32),
               [1, "a", None, (2.0, np.ones(3))], (np.zeros((2, 2)), "note")]
    for obj in samples:
        back = D.decode(D.encode(obj))
        if isinstance(obj, np.ndarray):
            assert back.dtype == obj.dtype and back.shape == obj.shape and np.array_equal(back, obj)
        elif isinstance(obj, (list, tuple)):
            assert type(back) is type(obj) and len(back) == len(obj)
        else:
            assert back == obj and type(back) is type(obj)
    nested = D.decode(D.encode([1, "a", None, (2.0, np.ones(3))]))
    assert nested[3][0] == 2.0 and np.array_equal(nested[3][1], np.ones(3))
    # only plain data travels: objects, callables, exotic dtypes are refused at the sender ...
    for bad in (object(), lambda: 0, {"a": 1}, np.array(["x"]), np.ones(2, dtype=np.complex128)):
        with pytest.raises(TypeError):
            D.encode(bad)
    # ... and a hostile or damaged message is a ValueError at the receiver, never code execution
    import pickle
    for blob in (pickle.dumps(os.system), b"", b"\x63", D.encode(np.ones(4))[:-3], D.encode("abc") + b"x",
                 bytes([7, 9, 1]) + b"\x00" * 8, bytes([8]) + (2 ** 60).to_bytes(8, "little")):
        with pytest.raises(ValueError):
            D.decode(blob)


def _socket_rank(rank, world, addresses, token, q):
    sys.path.insert(0, ROOT)
    from mellon_amd import distributed as D
    try:
        comm = D.SocketHostComm(addresses, rank, world, timeout=30.0, token=token)
        got = comm.allgather((rank, np.full(3, float(rank))))
        b = comm.broadcast(b"id-" + bytes([65 + rank]) if rank == 0 else None, src=0)
        comm.barrier()
        comm.close()
        q.put((rank, [g[0] for g in got], float(sum(g[1].sum() for g in got)), b))
    except BaseException as e:      # noqa: BLE001
        q.put((rank, "error", repr(e), None))


@pytest.mark.parametrize("transport", ["unix", "tcp-fallback"])
def test_socket_host_comm_handshake(transport):
    """Three processes over the host communicator: all-gather / broadcast / barrier; with `tcp-fallback` the Unix
    socket name is unusable for the non-zero ranks (they are given a different one), so they meet rank 0 on TCP.
    Meanwhile strangers knock: a wrong token, an out-of-range rank, garbage -- none of them takes a seat."""
    import multiprocessing as mp
    import socket as sk
    import struct
    import time
    ctx = mp.get_context("spawn")
    port = _free_port()
    name = f"mellon_amd.test.{os.getpid()}.{port}"
    addr0 = [("unix", name), ("tcp", "127.0.0.1", port)]
    addr_others = addr0 if transport == "unix" else [("unix", name + ".nobody-listens"), ("tcp", "127.0.0.1", port)]
    q = ctx.Queue()
    token = b"job-42"
    procs = [ctx.Process(target=_socket_rank, args=(0, 3, addr0, token, q))]
    procs[0].start()
    # strangers, before the real ranks arrive
    deadline = time.time() + 20
    knocked = 0
    while knocked < 3 and time.time() < deadline:
        try:
            s = sk.socket(sk.AF_INET, sk.SOCK_STREAM)
            s.settimeout(2.0)
            s.connect(("127.0.0.1", port))
            hello = b"MLNHC1" + bytes([len(token)]) + token
            payload = [b"MLNHC1" + bytes([5]) + b"wrong" + struct.pack("<q", 1),     # wrong token
                       hello + struct.pack("<q", 7),                                   # rank out of range
                       b"\x80\x04garbage-that-is-not-a-handshake-at-all........"][knocked]
            s.sendall(payload)
            try:
                assert s.recv(2) != b"ok"
            except (sk.timeout, ConnectionError):
                pass
            s.close()
            knocked += 1
        except (ConnectionRefusedError, OSError):
            time.sleep(0.05)
    assert knocked == 3
    for r in (1, 2):
        procs.append(ctx.Process(target=_socket_rank, args=(r, 3, addr_others, token, q)))
        procs[-1].start()
    results = sorted(q.get(timeout=60) for _ in range(3))
    for p in procs:
        p.join(timeout=30)
    for rank, seats, total, b in results:
        assert seats == [0, 1, 2], (rank, seats, total)
        assert total == 9.0 and b == b"id-A"
