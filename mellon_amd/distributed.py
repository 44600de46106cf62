"""Cell-sharded multi-GPU execution: one process per GPU, RCCL over xGMI inside one node.

Cells (rows of x, and with them the rows of L, nn_distances, V, Vdr, log_density_x) are split
into contiguous blocks, one per rank; landmarks, Lp, z and the predictor weights are replicated.
The reference has no distributed code at all -- the exchange steps follow from
the maths of the path:

  * per objective evaluation: all-reduce(sum) of [loss, grad] (m + 1 fp64)     inside mln_objective
  * once per fit: all-reduce(sum) of the m x m Ridge Gram and of L^T t         inside mln_ridge_init
  * heuristics: global mean of log nn (ls) and global 1 % quantile of mle (mu)  here, on the host

`Communicator` is the small interface the host logic needs; `RcclCommunicator` runs it through
the C-ABI (mln_comm_*), `TorchCommunicator` through torch.distributed (gloo) -- used to bootstrap
the RCCL unique id under torchrun and by the world_size-2 CPU tests.
"""
import os

import numpy as np


def shard_bounds(n, world_size, rank):
    """Contiguous [start, stop) row block of `rank`; sizes differ by at most one row."""
    base, rem = divmod(int(n), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class Communicator:
    rank = 0
    world_size = 1

    def allreduce_sum(self, a):
        return np.asarray(a, dtype=np.float64)

    def allgather_rows(self, a, counts=None):
        """Concatenate the ranks' 1-D arrays in rank order."""
        return np.asarray(a, dtype=np.float64)

    def barrier(self):
        pass

    # -- reductions the heuristics need (parameters.py:599,613 on the GLOBAL cell set) -----------------
    def global_mean(self, a):
        a = np.asarray(a, dtype=np.float64)
        s = self.allreduce_sum(np.array([a.sum(), float(a.size)]))
        return float(s[0] / s[1])

    def global_quantile(self, a, q):
        return float(np.quantile(self.allgather_rows(a), q))


class TorchCommunicator(Communicator):
    """torch.distributed (gloo on CPU) -- bootstrap and CPU tests only; no GPU data path."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist, self._group = dist, group
        self.rank, self.world_size = dist.get_rank(group), dist.get_world_size(group)

    def allreduce_sum(self, a):
        import torch
        t = torch.from_numpy(np.array(a, dtype=np.float64, copy=True))
        self._dist.all_reduce(t, group=self._group)
        return t.numpy()

    def allgather_rows(self, a, counts=None):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
        sizes = self.allreduce_sum(np.eye(self.world_size)[self.rank] * a.size).astype(np.int64)
        offs = np.concatenate([[0], np.cumsum(sizes)])
        buf = np.zeros(int(offs[-1]))
        buf[offs[self.rank]:offs[self.rank + 1]] = a
        return self.allreduce_sum(buf)

    def barrier(self):
        self._dist.barrier(group=self._group)

    def broadcast_bytes(self, payload, src=0):
        import torch
        n = len(payload) if self.rank == src else 0
        size = torch.tensor([n], dtype=torch.int64)
        self._dist.broadcast(size, src, group=self._group)
        t = torch.zeros(int(size.item()), dtype=torch.uint8)
        if self.rank == src:
            t = torch.tensor(list(payload), dtype=torch.uint8)
        self._dist.broadcast(t, src, group=self._group)
        return bytes(t.tolist())


class RcclCommunicator(Communicator):
    """Collectives through libmellon_hip.so (RCCL over xGMI)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.rank, self.world_size = ctx.rank, ctx.n_ranks

    def allreduce_sum(self, a):
        return self.ctx.allreduce_sum(np.asarray(a, dtype=np.float64))

    def allgather_rows(self, a, counts=None):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
        sizes = self.allreduce_sum(np.eye(self.world_size)[self.rank] * a.size).astype(np.int64)
        offs = np.concatenate([[0], np.cumsum(sizes)])
        buf = np.zeros(int(offs[-1]))
        buf[offs[self.rank]:offs[self.rank + 1]] = a
        return self.allreduce_sum(buf)

    def barrier(self):
        self.allreduce_sum(np.zeros(1))


_current = Communicator()


def current():
    return _current


def set_current(comm):
    global _current
    _current = comm
    return comm


def init_from_env(backend="gloo"):
    """Under `python -m torch.distributed.run`: bootstrap RCCL on GPU LOCAL_RANK.

    torch.distributed (gloo, 127.0.0.1 rendezvous from MASTER_ADDR/PORT) only carries the 128-byte
    RCCL unique id and barriers; every data-path collective runs in libmellon_hip.so."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force = os.environ.get("MELLON_AMD_FORCE_COMM") == "1" and "RANK" in os.environ   # 1-rank RCCL (testing)
    if world <= 1 and not force:
        return set_current(Communicator())
    import torch.distributed as dist
    from . import _lib
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
    boot = TorchCommunicator()
    ctx = _lib.default_context()                       # device = LOCAL_RANK
    uid = ctx.comm_unique_id() if boot.rank == 0 else b""
    uid = boot.broadcast_bytes(uid, src=0)
    ctx.comm_init(uid, boot.world_size, boot.rank)
    comm = RcclCommunicator(ctx)
    comm.bootstrap = boot
    return set_current(comm)
