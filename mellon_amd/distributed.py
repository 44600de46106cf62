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
import sys
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

    def reduce_sum(self, a):
        """Sum of the ranks' float64 arrays, added in rank order (the same bits on every rank)."""
        parts = self.allgather(np.ascontiguousarray(a, dtype=np.float64))
        out = np.array(parts[0], dtype=np.float64)
        for p in parts[1:]:
            out = out + p
        return out

    def close(self):
        pass


# Wire format of the host communicator.  Only plain data ever travels (rank ids, counts, float64 vectors of the
# heuristics, the 128-byte RCCL id, landmark matrices, short notes), so the codec knows exactly these types and
# nothing else: a peer can make a rank allocate memory, never run code (no pickle on a socket any local process
# can connect to).  One value = 1 tag byte + payload; containers nest.
_T_NONE, _T_FALSE, _T_TRUE, _T_INT, _T_FLOAT, _T_STR, _T_BYTES, _T_ARRAY, _T_LIST, _T_TUPLE = range(10)
_DTYPES = ("float64", "float32", "int64", "int32", "uint8", "bool")
_MAX_MSG = 1 << 34        # 16 GiB: nothing legitimate is larger than a landmark matrix
_MAX_DEPTH = 8


def _encode(obj, out, depth=0):
    if depth > _MAX_DEPTH:
        raise TypeError("host communicator: value nests too deeply")
    if obj is None:
        out.append(bytes([_T_NONE]))
    elif obj is True or obj is False or isinstance(obj, np.bool_):
        out.append(bytes([_T_TRUE if obj else _T_FALSE]))
    elif isinstance(obj, (int, np.integer)):
        out.append(bytes([_T_INT]) + struct.pack("<q", int(obj)))
    elif isinstance(obj, (float, np.floating)):
        out.append(bytes([_T_FLOAT]) + struct.pack("<d", float(obj)))
    elif isinstance(obj, str):
        b = obj.encode("utf-8")
        out.append(bytes([_T_STR]) + struct.pack("<Q", len(b)) + b)
    elif isinstance(obj, (bytes, bytearray, memoryview)):
        b = bytes(obj)
        out.append(bytes([_T_BYTES]) + struct.pack("<Q", len(b)) + b)
    elif isinstance(obj, np.ndarray):
        name = obj.dtype.name
        if name not in _DTYPES:
            raise TypeError(f"host communicator: arrays of dtype {name} do not travel")
        a = np.ascontiguousarray(obj)
        head = bytes([_T_ARRAY, _DTYPES.index(name), a.ndim]) + struct.pack(f"<{a.ndim}q", *a.shape)
        out.append(head)
        out.append(a.tobytes())
    elif isinstance(obj, (list, tuple)):
        out.append(bytes([_T_LIST if isinstance(obj, list) else _T_TUPLE]) + struct.pack("<Q", len(obj)))
        for item in obj:
            _encode(item, out, depth + 1)
    else:
        raise TypeError(f"host communicator: values of type {type(obj).__name__} do not travel")


def encode(obj):
    out = []
    _encode(obj, out)
    return b"".join(out)


def _decode(buf, pos, depth=0):
    if depth > _MAX_DEPTH or pos >= len(buf):
        raise ValueError("host communicator: malformed message")
    tag = buf[pos]
    pos += 1

    def take(k):
        nonlocal pos
        if k < 0 or pos + k > len(buf):
            raise ValueError("host communicator: truncated message")
        piece = buf[pos:pos + k]
        pos += k
        return piece

    if tag == _T_NONE:
        return None, pos
    if tag in (_T_FALSE, _T_TRUE):
        return tag == _T_TRUE, pos
    if tag == _T_INT:
        return struct.unpack("<q", take(8))[0], pos
    if tag == _T_FLOAT:
        return struct.unpack("<d", take(8))[0], pos
    if tag in (_T_STR, _T_BYTES):
        (k,) = struct.unpack("<Q", take(8))
        b = bytes(take(k))
        return (b.decode("utf-8") if tag == _T_STR else b), pos
    if tag == _T_ARRAY:
        code, ndim = take(2)
        if code >= len(_DTYPES) or ndim > 8:
            raise ValueError("host communicator: malformed array header")
        shape = struct.unpack(f"<{ndim}q", take(8 * ndim))
        if any(k < 0 for k in shape):
            raise ValueError("host communicator: negative array extent")
        dt = np.dtype(_DTYPES[code])
        count = 1
        for k in shape:
            count *= k
        raw = take(count * dt.itemsize)
        return np.frombuffer(raw, dtype=dt).reshape(shape).copy(), pos
    if tag in (_T_LIST, _T_TUPLE):
        (k,) = struct.unpack("<Q", take(8))
        if k > len(buf):
            raise ValueError("host communicator: malformed container")
        items = []
        for _ in range(k):
            item, pos = _decode(buf, pos, depth + 1)
            items.append(item)
        return (items if tag == _T_LIST else tuple(items)), pos
    raise ValueError(f"host communicator: unknown type tag {tag}")


def decode(buf):
    obj, pos = _decode(memoryview(buf), 0)
    if pos != len(buf):
        raise ValueError("host communicator: trailing bytes in message")
    return obj


_MAGIC = b"MLNHC1"


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


def _recv_msg(sock, limit=_MAX_MSG):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > limit:
        raise ValueError(f"host communicator: message of {n} bytes refused")
    return _recv_exact(sock, n)


def _describe_env():
    keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MELLON_AMD_PORT")
    return ", ".join(f"{k}={os.environ.get(k, '<unset>')}" for k in keys)


class SocketHostComm(HostComm):
    """Star all-gather over stream sockets: rank 0 listens, collects one message per rank and sends
    the list back.  `address` is ("unix", name) -- an abstract-namespace Unix socket, the default on
    one node: no port to collide with, gone when the processes exit -- or ("tcp", host, port).
    A list of addresses is tried in order (rank 0 listens on all of them): the Unix socket first, TCP on
    MASTER_ADDR as the fallback when the ranks do not share a network namespace.

    Messages use the typed binary codec above (no pickle).  The handshake is `magic | job token | rank`; a
    connection with the wrong token, a rank outside [1, world_size) or a rank already seated is dropped."""

    def __init__(self, address, rank, world_size, timeout=300.0, token=b""):
        self.rank, self.world_size = int(rank), int(world_size)
        self._peers = []
        self._sock = None
        addresses = [address] if isinstance(address[0], str) else list(address)
        token = bytes(token)[:64]
        hello = _MAGIC + struct.pack("<B", len(token)) + token

        def open_socket(addr):
            kind = addr[0]
            fam = socket.AF_UNIX if kind == "unix" else socket.AF_INET
            target = ("\0" + addr[1]) if kind == "unix" else (addr[1], int(addr[2]))
            return kind, socket.socket(fam, socket.SOCK_STREAM), target

        if self.rank == 0:
            servers = []
            for addr in addresses:
                kind, srv, target = open_socket(addr)
                try:
                    if kind == "tcp":
                        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind(target)
                    srv.listen(self.world_size)
                    srv.setblocking(False)
                    servers.append((kind, srv))
                except OSError:
                    srv.close()
            if not servers:
                raise OSError(f"rank 0: cannot listen on any of {addresses} ({_describe_env()})")
            import select
            peers = {}
            deadline = time.time() + timeout
            while len(peers) < self.world_size - 1:
                left = deadline - time.time()
                if left <= 0:
                    missing = sorted(set(range(1, self.world_size)) - set(peers))
                    for _, srv in servers:
                        srv.close()
                    raise TimeoutError(f"rank 0: ranks {missing} never reached the host communicator at {addresses} "
                                       f"within {timeout:.0f} s ({_describe_env()})")
                ready, _, _ = select.select([srv for _, srv in servers], [], [], min(left, 1.0))
                for srv in ready:
                    try:
                        conn, _ = srv.accept()
                    except OSError:
                        continue
                    conn.settimeout(10.0)
                    try:
                        head = _recv_exact(conn, len(hello) + 8)
                        (r,) = struct.unpack("<q", head[len(hello):])
                        if head[:len(hello)] != hello or not (1 <= r < self.world_size) or r in peers:
                            raise ValueError("handshake refused")
                    except (ValueError, OSError, ConnectionError):
                        conn.close()         # not one of ours (or a duplicate): the seat stays free
                        continue
                    conn.settimeout(timeout)
                    if conn.family == socket.AF_INET:
                        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    conn.sendall(b"ok")
                    peers[int(r)] = conn
            for _, srv in servers:
                srv.close()
            self._peers = [peers[r] for r in range(1, self.world_size)]
        else:
            deadline = time.time() + timeout
            s = None
            while s is None:
                for addr in addresses:
                    kind, cand, target = open_socket(addr)
                    try:
                        cand.settimeout(5.0)
                        cand.connect(target)
                        if kind == "tcp":
                            cand.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        cand.sendall(hello + struct.pack("<q", self.rank))
                        if _recv_exact(cand, 2) != b"ok":
                            raise ConnectionError("handshake refused")
                        s = cand
                        break
                    except (OSError, ConnectionError):
                        cand.close()
                if s is None:
                    if time.time() > deadline:
                        raise TimeoutError(f"rank {self.rank}: cannot reach the host communicator of rank 0 at {addresses} "
                                           f"within {timeout:.0f} s ({_describe_env()})")
                    time.sleep(0.05)
            s.settimeout(timeout)
            self._sock = s

    def allgather(self, obj):
        if self.world_size == 1:
            return [obj]
        if self.rank == 0:
            items = [obj] + [decode(_recv_msg(c)) for c in self._peers]
            blob = encode(items)
            for c in self._peers:
                _send_msg(c, blob)
            return items
        _send_msg(self._sock, encode(obj))
        return decode(_recv_msg(self._sock))

    def reduce_sum(self, a):
        """Star reduction: rank 0 adds the peers' arrays in rank order and returns the sum to everyone -- O(world) array
        transfers where the all-gather form needs O(world^2) (the host-staged device collectives move 100 MB sums)."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        if self.world_size == 1:
            return a.copy()
        if self.rank == 0:
            out = a.copy()
            for c in self._peers:
                part = decode(_recv_msg(c))
                if part.shape != out.shape:
                    raise ValueError(f"reduce_sum: a peer sent shape {part.shape}, expected {out.shape}")
                out += part
            blob = encode(out)
            for c in self._peers:
                _send_msg(c, blob)
            return out
        _send_msg(self._sock, encode(a))
        return decode(_recv_msg(self._sock))

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


def _host_address(channel=0):
    """Where the ranks' host sides meet: MELLON_AMD_PORT set -> TCP only; else the abstract Unix socket named after
    the launcher's rendezvous, with TCP on MASTER_ADDR : MASTER_PORT + 1 as the fallback (tried by every rank in
    the same order; rank 0 listens on both).  channel 1: the second connection that carries host-staged DEVICE
    collectives (own sockets: a worker thread may be talking on channel 0 at the same time)."""
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = os.environ.get("MELLON_AMD_PORT")
    if port:
        return [("tcp", addr, int(port) + 2 * channel)]
    run = os.environ.get("TORCHELASTIC_RUN_ID", "none")
    out = [("unix", f"mellon_amd.{addr}.{os.environ.get('MASTER_PORT', '0')}.{run}" + (f".ch{channel}" if channel else ""))]
    try:
        mp = int(os.environ.get("MASTER_PORT", "0"))
    except ValueError:
        mp = 0
    if 0 < mp < 65533:
        out.append(("tcp", addr, mp + 1 + 2 * channel))
    return out


def _job_token():
    """Shared by the ranks of one launch and by nobody who cannot read their environment."""
    tok = os.environ.get("MELLON_AMD_TOKEN") or os.environ.get("TORCHELASTIC_RUN_ID", "")
    return tok.encode("utf-8")[:64]


def _with_deadline(what, fn, timeout):
    """Run `fn()` (a call into the C library: ctypes releases the GIL) and give up with a readable error when it
    does not return -- a communicator whose peers never arrive blocks inside RCCL, where no exception can reach."""
    box = {}

    def body():
        try:
            box["value"] = fn()
        except BaseException as e:      # noqa: BLE001 -- re-raised on the caller's thread
            box["error"] = e

    state = thread_state()
    t = threading.Thread(target=lambda: (adopt_thread_state(state), body()), daemon=True)
    t.start()
    t.join(timeout)
    if t.is_alive():
        raise TimeoutError(f"{what} did not complete within {timeout:.0f} s ({_describe_env()}; visible GPUs: "
                           f"{_visible_devices()}).  Every rank must reach this point: check that WORLD_SIZE processes "
                           "were started, that LOCAL_RANK indexes a visible GPU, and that HSA_ENABLE_IPC_MODE_LEGACY=0 "
                           "is exported (RCCL's intra-node transport needs dmabuf IPC on this driver).")
    if "error" in box:
        raise box["error"]
    return box.get("value")


def _visible_devices():
    try:
        from . import _lib
        return _lib.device_count()
    except Exception:      # noqa: BLE001 -- diagnostic text only
        return "unknown"


def self_test(comm, timeout=120.0):
    """First collectives of a fresh communicator, checked: all-reduce of (rank + 1) and of a rank-dependent vector,
    and the host-side exchange.  Returns a small report; raises with the environment spelled out on any mismatch."""
    w, r = comm.world_size, comm.rank
    if w == 1 and comm.ctx is None:
        return {"world_size": 1, "ok": True}
    t0 = time.perf_counter()
    probe = np.concatenate([[r + 1.0], np.arange(1024, dtype=np.float64) * (r + 1)])
    got = _with_deadline("the first device all-reduce (RCCL)", lambda: comm.ctx.allreduce_sum(probe), timeout)
    tri = w * (w + 1) / 2.0
    want = np.concatenate([[tri], np.arange(1024, dtype=np.float64) * tri])
    if got.shape != want.shape or not np.array_equal(got, want):
        raise RuntimeError(f"rank {r}: device all-reduce self-test failed (sum of rank ids {got[0]} instead of {tri}; "
                           f"{_describe_env()})")
    dt_dev = time.perf_counter() - t0
    ranks = comm.host.allgather(r) if hasattr(comm, "host") else [r]
    if list(ranks) != list(range(w)):
        raise RuntimeError(f"rank {r}: host communicator seats are {ranks}, expected 0..{w - 1} ({_describe_env()})")
    sizes = [1 << 10, 1 << 20]
    lat = []
    for count in sizes:      # two sizes: latency of the per-evaluation all-reduce, bandwidth of the Gram's
        buf = np.ones(count // 8)
        comm.ctx.allreduce_sum(buf)
        t1 = time.perf_counter()
        for _ in range(5):
            comm.ctx.allreduce_sum(buf)
        lat.append((time.perf_counter() - t1) / 5)
    return {"world_size": w, "ok": True, "first_allreduce_s": dt_dev,
            "allreduce_host_buffers_s": {str(b): t for b, t in zip(sizes, lat)}}


def init_from_env(self_check=True):
    """One process per GPU (launched e.g. by `python -m torch.distributed.run`): read RANK / WORLD_SIZE /
    LOCAL_RANK / MASTER_*, connect the host communicator, create the RCCL communicator on GPU LOCAL_RANK and run
    its self-test (`self_test`: the first collectives, checked, under a deadline)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force = os.environ.get("MELLON_AMD_FORCE_COMM") == "1" and "RANK" in os.environ   # 1-rank RCCL (testing)
    if world <= 1 and not force:
        return set_current(Communicator())
    from . import _lib
    rank = int(os.environ.get("RANK", "0"))
    if not 0 <= rank < world:
        raise ValueError(f"RANK={rank} outside [0, WORLD_SIZE={world}) ({_describe_env()})")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = _lib.device_count()
    backend = os.environ.get("MELLON_AMD_COMM", "rccl").lower()
    if backend not in ("rccl", "host"):
        raise ValueError(f"MELLON_AMD_COMM={backend!r}: expected 'rccl' or 'host'")
    if os.environ.get("MELLON_AMD_SHARE_GPU") == "1" and n_dev > 0:
        # several ranks per GPU (testing the multi-process path on a small box): RCCL refuses that, host-staged works
        os.environ["MELLON_AMD_DEVICE"] = str(local % n_dev)
        backend = "host"
    elif local >= n_dev:
        raise RuntimeError(f"LOCAL_RANK={local} but only {n_dev} GPU(s) are visible to this process ({_describe_env()}; "
                           "HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES restrict the set; MELLON_AMD_SHARE_GPU=1 lets "
                           "ranks share a GPU over host-staged collectives)")
    timeout = float(os.environ.get("MELLON_AMD_COMM_TIMEOUT", "300"))
    host = SocketHostComm(_host_address(), rank, world, timeout=timeout, token=_job_token())
    ctx = _lib.default_context()                       # device = LOCAL_RANK
    note = None
    if backend == "rccl":
        # An ERROR from the library (librccl missing, ncclCommInitRank refusing the topology) is survivable: if every rank
        # saw one, all of them move to host-staged collectives.  A HANG is not -- it runs into the deadline and raises.
        err, uid = None, None
        if rank == 0:
            try:
                uid = ctx.comm_unique_id()
            except _lib.MellonHipError as e:
                err = e
        uid = host.broadcast(uid, src=0)              # None: rank 0 could not even load RCCL
        if uid is None:
            err = err or RuntimeError("rank 0 could not create the RCCL id")
        else:
            try:
                _with_deadline("RCCL communicator set-up (ncclCommInitRank)", lambda: ctx.comm_init(uid, world, rank), timeout)
            except _lib.MellonHipError as e:
                err = e
        errs = host.allgather(None if err is None else str(err))
        if any(e is not None for e in errs):
            if not all(e is not None for e in errs):
                raise RuntimeError(f"rank {rank}: RCCL set-up failed on ranks {[i for i, e in enumerate(errs) if e is not None]} "
                                   f"only ({next(e for e in errs if e is not None)}); {_describe_env()}")
            note = f"RCCL unavailable ({errs[0]}): host-staged collectives"
            if rank == 0:
                print(f"[mellon_amd] {note}", file=sys.stderr)
            backend = "host"
    if backend == "host":
        staged = HostStagedCollectives(SocketHostComm(_host_address(channel=1), rank, world, timeout=timeout, token=_job_token()))
        ctx.comm_init_host(world, rank, staged)
    comm = set_current(ShardedCommunicator(ctx, host))
    comm.backend, comm.backend_note = backend, note
    if self_check:
        comm.self_test_report = self_test(comm, timeout=min(timeout, 120.0))
        comm.self_test_report["backend"] = backend
    return comm


class HostStagedCollectives:
    """The callback behind mln_comm_init_host: device collectives staged through host memory and carried by a host
    communicator of their own (include/mellon_hip.h: op 0 all-reduce, 1 broadcast from rank 0, 2 all-gather)."""

    def __init__(self, host):
        self.host = host
        self.failure = None

    def __call__(self, user, op, buf, buf2, count):
        import ctypes
        try:
            a = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ctypes.c_double)), shape=(int(count),))
            if op == 0:
                a[:] = self.host.reduce_sum(a)
            elif op == 1:
                a[:] = self.host.broadcast(a.copy() if self.host.rank == 0 else None, src=0)
            elif op == 2:
                out = np.ctypeslib.as_array(ctypes.cast(buf2, ctypes.POINTER(ctypes.c_double)),
                                            shape=(int(count) * self.host.world_size,))
                out[:] = np.concatenate(self.host.allgather(a.copy()))
            else:
                raise ValueError(f"unknown collective {op}")
            return 0
        except BaseException as e:          # noqa: BLE001 -- must not unwind through the C frame
            self.failure = e
            return 1


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
