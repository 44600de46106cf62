"""Cell-sharded multi-GPU execution: one process per GPU, RCCL over xGMI inside one node.

Cells (rows of x, and with them the rows of K / L, nn_distances, V, Vdr, log_density_x) are split
into contiguous blocks, one per rank; landmarks, Lp, z and the predictor weights are replicated.
The reference has no distributed code at all -- the exchange steps follow from
the maths of the path:

  * per objective evaluation: all-reduce(sum) of [loss, grad] (m + 1 fp64)     inside libmellon_hip.so
  * once per fit: all-reduce(sum) of the m x m Ridge Gram and of L^T t         inside libmellon_hip.so
  * heuristics: global mean of log nn (ls) and global 1 % quantile of mle (mu)  here, on the host

Two layers:

  * the DEVICE collectives live in the C library (csrc/comm.hip): RCCL, or the in-process loopback
    group that runs N ranks as N host threads on one GPU;
  * the HOST side only moves a few kilobytes (the 128-byte RCCL id, scalar heuristics, barriers) and
    does so over a plain socket (`SocketHostComm`, standard library only) or, for thread-ranks,
    through shared memory (`ThreadHostComm`).  No framework is involved: under
    `python -m torch.distributed.run` only the environment variables RANK / WORLD_SIZE /
    LOCAL_RANK / MASTER_ADDR / MASTER_PORT are read.
"""
import os
import pickle
import socket
import struct
import threading
import time

import numpy as np


def shard_bounds(n, world_size, rank):
    """Contiguous [start, stop) row block of `rank`; sizes differ by at most one row."""
    base, rem = divmod(int(n), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


# ---- host-side exchange ---------------------------------------------------------------------------------
class HostComm:
    """allgather of small Python objects between the ranks' host processes / threads."""
    rank = 0
    world_size = 1

    def allgather(self, obj):
        return [obj]

    def barrier(self):
        self.allgather(None)

    def broadcast(self, obj, src=0):
        return self.allgather(obj if self.rank == src else None)[src]

    def close(self):
        pass


def _send_msg(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("peer closed the host communicator socket")
        buf += chunk
    return bytes(buf)


def _recv_msg(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class SocketHostComm(HostComm):
    """Star all-gather over stream sockets: rank 0 listens, collects one message per rank and sends
    the list back.  `address` is ("unix", name) -- an abstract-namespace Unix socket, the default on
    one node: no port to collide with, gone when the processes exit -- or ("tcp", host, port)."""

    def __init__(self, address, rank, world_size, timeout=300.0):
        self.rank, self.world_size = int(rank), int(world_size)
        self._peers = []
        self._sock = None
        kind = address[0]
        fam = socket.AF_UNIX if kind == "unix" else socket.AF_INET
        target = ("\0" + address[1]) if kind == "unix" else (address[1], int(address[2]))
        if self.rank == 0:
            srv = socket.socket(fam, socket.SOCK_STREAM)
            if kind == "tcp":
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(target)
            srv.listen(self.world_size)
            srv.settimeout(timeout)
            peers = {}
            while len(peers) < self.world_size - 1:
                conn, _ = srv.accept()
                conn.settimeout(timeout)
                if kind == "tcp":
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                r = pickle.loads(_recv_msg(conn))
                peers[int(r)] = conn
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world_size)]
        else:
            deadline = time.time() + timeout
            while True:
                s = socket.socket(fam, socket.SOCK_STREAM)
                try:
                    s.connect(target)
                    break
                except (ConnectionRefusedError, FileNotFoundError, OSError):
                    s.close()
                    if time.time() > deadline:
                        raise TimeoutError(f"rank {self.rank}: cannot reach the host communicator of rank 0 at {address}")
                    time.sleep(0.02)
            s.settimeout(timeout)
            if kind == "tcp":
                s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            _send_msg(s, pickle.dumps(self.rank))
            self._sock = s

    def allgather(self, obj):
        if self.world_size == 1:
            return [obj]
        if self.rank == 0:
            items = [obj] + [pickle.loads(_recv_msg(c)) for c in self._peers]
            blob = pickle.dumps(items, protocol=pickle.HIGHEST_PROTOCOL)
            for c in self._peers:
                _send_msg(c, blob)
            return items
        _send_msg(self._sock, pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL))
        return pickle.loads(_recv_msg(self._sock))

    def close(self):
        for c in self._peers:
            c.close()
        if self._sock is not None:
            self._sock.close()
        self._peers, self._sock = [], None


class ThreadGroup:
    """Shared state of the thread-ranks of one process (host side of the loopback communicator)."""

    def __init__(self, world_size):
        self.world_size = int(world_size)
        self.slots = [None] * self.world_size
        self.barrier = threading.Barrier(self.world_size)


class ThreadHostComm(HostComm):
    def __init__(self, group, rank):
        self.group, self.rank, self.world_size = group, int(rank), group.world_size

    def allgather(self, obj):
        g = self.group
        g.slots[self.rank] = obj
        g.barrier.wait()
        items = list(g.slots)
        g.barrier.wait()          # nobody overwrites a slot before everyone has read it
        return items


# ---- what the host logic sees -----------------------------------------------------------------------------
class Communicator:
    """Single rank: every reduction is the identity."""
    rank = 0
    world_size = 1
    ctx = None        # device context bound to this rank (None: the process-wide default)

    def allreduce_sum(self, a):
        return np.array(a, dtype=np.float64)

    def allgather_rows(self, a):
        """Concatenate the ranks' 1-D arrays in rank order."""
        return np.asarray(a, dtype=np.float64).reshape(-1)

    def broadcast(self, obj, src=0):
        return obj

    def barrier(self):
        pass

    # -- reductions the heuristics need (parameters.py:599,613 on the GLOBAL cell set) -----------------
    def global_count(self, n_local):
        return int(n_local)

    def global_offset(self, n_local):
        """(global index of this rank's first cell, total number of cells) for contiguous shards in rank order."""
        return 0, int(n_local)

    def global_mean(self, a):
        a = np.asarray(a, dtype=np.float64)
        s = self.allreduce_sum(np.array([a.sum(), float(a.size)]))
        return float(s[0] / s[1])

    def global_quantile(self, a, q):
        return float(np.quantile(np.asarray(a, dtype=np.float64), q))


class ShardedCommunicator(Communicator):
    """N ranks.  Host-side reductions travel over `host` (a HostComm) and are summed in rank order on
    every rank -- identical bits everywhere, no device involvement, so they may run next to device work
    of the same rank; the data-path collectives are issued by libmellon_hip.so on `ctx`."""

    def __init__(self, ctx, host):
        self.ctx, self.host = ctx, host
        self.rank, self.world_size = host.rank, host.world_size

    def allreduce_sum(self, a):
        a = np.array(a, dtype=np.float64)
        parts = self.host.allgather(a)
        out = np.array(parts[0], dtype=np.float64)
        for p in parts[1:]:
            out = out + p
        return out

    def allgather_rows(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
        return np.concatenate(self.host.allgather(a))

    def broadcast(self, obj, src=0):
        return self.host.broadcast(obj, src)

    def barrier(self):
        self.host.barrier()

    def global_count(self, n_local):
        return int(sum(self.host.allgather(int(n_local))))

    def global_offset(self, n_local):
        counts = self.host.allgather(int(n_local))
        return int(sum(counts[:self.rank])), int(sum(counts))

    def global_quantile(self, a, q):
        """np.quantile(all cells, q) (linear interpolation, jnp.quantile's default) without gathering all
        cells: the order statistics needed are among each rank's own smallest (or largest) few."""
        a = np.asarray(a, dtype=np.float64).reshape(-1)
        n = self.global_count(a.size)
        pos = q * (n - 1)
        lo = int(np.floor(pos))
        hi = min(lo + 1, n - 1)
        if hi + 1 <= n - lo:            # the two order statistics are among the hi + 1 smallest values
            k = min(hi + 1, a.size)
            cand = np.partition(a, k - 1)[:k] if k < a.size else a
            allc = np.sort(np.concatenate(self.host.allgather(cand)))
            v_lo, v_hi = allc[lo], allc[hi]
        else:                           # ... or among the n - lo largest
            k = min(n - lo, a.size)
            cand = np.partition(a, a.size - k)[a.size - k:] if k < a.size else a
            allc = np.sort(np.concatenate(self.host.allgather(cand)))
            v_lo, v_hi = allc[lo - (n - allc.size)], allc[hi - (n - allc.size)]
        return float(v_lo + (v_hi - v_lo) * (pos - lo))


_current = Communicator()
_tls = threading.local()


def current():
    """The communicator of the calling thread (thread-ranks), else the process-wide one."""
    return getattr(_tls, "comm", None) or _current


def set_current(comm):
    global _current
    _current = comm
    return comm


def set_thread_current(comm):
    """Bind `comm` (and its device context) to the calling thread; None unbinds."""
    _tls.comm = comm
    return comm


def thread_state():
    """What a helper thread must inherit to act for the same rank (see adopt_thread_state)."""
    return getattr(_tls, "comm", None)


def adopt_thread_state(state):
    _tls.comm = state


def _host_address():
    port = os.environ.get("MELLON_AMD_PORT")
    if port:
        return ("tcp", os.environ.get("MASTER_ADDR", "127.0.0.1"), int(port))
    run = os.environ.get("TORCHELASTIC_RUN_ID", "none")
    return ("unix", f"mellon_amd.{os.environ.get('MASTER_ADDR', '127.0.0.1')}.{os.environ.get('MASTER_PORT', '0')}.{run}")


def init_from_env():
    """One process per GPU (launched e.g. by `python -m torch.distributed.run`): read RANK / WORLD_SIZE /
    LOCAL_RANK / MASTER_*, connect the host communicator, create the RCCL communicator on GPU LOCAL_RANK."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force = os.environ.get("MELLON_AMD_FORCE_COMM") == "1" and "RANK" in os.environ   # 1-rank RCCL (testing)
    if world <= 1 and not force:
        return set_current(Communicator())
    from . import _lib
    rank = int(os.environ.get("RANK", "0"))
    host = SocketHostComm(_host_address(), rank, world)
    ctx = _lib.default_context()                       # device = LOCAL_RANK
    uid = host.broadcast(ctx.comm_unique_id() if rank == 0 else None, src=0)
    ctx.comm_init(uid, world, rank)
    return set_current(ShardedCommunicator(ctx, host))


def run_loopback(n_ranks, fn, device=None):
    """Run `fn(comm)` as n_ranks thread-ranks of this process, each with its own device context (all on
    GPU `device`) joined by the library's loopback communicator; returns the results in rank order.
    The sharded code path is exactly the multi-process one; only the transport differs."""
    from . import _lib
    group = _lib.LoopbackGroup(n_ranks)
    tgroup = ThreadGroup(n_ranks)
    results, errors = [None] * n_ranks, [None] * n_ranks

    def body(rank):
        ctx = None
        try:
            ctx = _lib.Context(device)
            ctx.comm_init_loopback(group, rank)
            comm = ShardedCommunicator(ctx, ThreadHostComm(tgroup, rank))
            set_thread_current(comm)
            results[rank] = fn(comm)
        except BaseException as e:      # noqa: BLE001 -- re-raised below
            errors[rank] = e
            tgroup.barrier.abort()
            group.abort()
        finally:
            set_thread_current(None)

    threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(n_ranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    def secondary(e):      # a rank that only noticed that a PEER had failed
        return isinstance(e, threading.BrokenBarrierError) or "a peer rank failed" in str(e)
    first = next((e for e in errors if e is not None and not secondary(e)), None)
    first = first or next((e for e in errors if e is not None), None)
    if first is not None:
        raise first
    return results
