"""ctypes binding of libmellon_hip.so (include/mellon_hip.h).

This is the whole Python <-> device boundary: NumPy float64 C-contiguous arrays (or
``DeviceArray`` handles) in, NumPy arrays out.  There is no CPU fallback -- if the shared
library is missing or no gfx950 device is visible, every compute entry raises.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmellon_hip.so")

MLN_OK, MLN_ERR_NOT_PD, MLN_ERR_SHAPE, MLN_ERR_HIP, MLN_ERR_RCCL, MLN_ERR_ARG, MLN_ERR_UNSUPPORTED = range(7)
MLN_UNIQUE_ID_BYTES = 128
MLN_N_STAGE_TIMES = 22

K_MATERN32, K_MATERN52, K_EXPQUAD, K_EXPONENTIAL, K_RATQUAD, K_LINEAR, K_DISTANCE = 1, 2, 3, 4, 5, 6, 7
OP_LEAF, OP_CONST, OP_ADD, OP_MUL, OP_POW = 0, 1, 2, 3, 4


HOST_COLLECTIVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64)


class MellonHipError(RuntimeError):
    """HIP / RCCL / argument failure inside libmellon_hip.so."""


class Leaf(C.Structure):
    _fields_ = [("kind", C.c_int32), ("ndims", C.c_int32), ("ls", C.c_double), ("alpha", C.c_double),
                ("dims", C.POINTER(C.c_int32))]


class Tok(C.Structure):
    _fields_ = [("op", C.c_int32), ("leaf", C.c_int32), ("value", C.c_double)]


class SolverOpts(C.Structure):
    _fields_ = [("maxiter", C.c_int32), ("maxcor", C.c_int32), ("maxls", C.c_int32), ("ftol", C.c_double),
                ("gtol", C.c_double)]


class KernelDesc(C.Structure):
    _fields_ = [("n_leaves", C.c_int32), ("n_toks", C.c_int32), ("leaves", C.POINTER(Leaf)),
                ("toks", C.POINTER(Tok))]


# every symbol include/mellon_hip.h declares: (name, restype, argtypes)
_vp, _i64, _i32, _dbl, _dp = C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_void_p
_KD = C.POINTER(KernelDesc)
SYMBOLS = [
    ("mln_ctx_create", C.c_int, [C.c_int, C.POINTER(_vp)]),
    ("mln_ctx_destroy", None, [_vp]),
    ("mln_last_error", C.c_char_p, [_vp]),
    ("mln_device_info", C.c_int, [_vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(_i64)]),
    ("mln_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("mln_synchronize", C.c_int, [_vp]),
    ("mln_malloc", C.c_int, [_vp, _i64, C.POINTER(_vp)]),
    ("mln_free", C.c_int, [_vp, _vp]),
    ("mln_memcpy", C.c_int, [_vp, _vp, _vp, _i64]),
    ("mln_host_register", C.c_int, [_vp, _vp, _i64]),
    ("mln_host_unregister", C.c_int, [_vp, _vp]),
    ("mln_release_cached_memory", C.c_int, []),
    ("mln_comm_unique_id", C.c_int, [_vp]),
    ("mln_comm_init", C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    ("mln_comm_allreduce_sum", C.c_int, [_vp, _dp, _i64]),
    ("mln_loopback_create", C.c_int, [C.c_int, C.POINTER(_vp)]),
    ("mln_loopback_destroy", None, [_vp]),
    ("mln_comm_init_loopback", C.c_int, [_vp, _vp, C.c_int]),
    ("mln_comm_init_host", C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    ("mln_loopback_abort", None, [_vp]),
    ("mln_comm_info", C.c_int, [_vp, C.POINTER(_i32), _dp, _i32]),
    ("mln_kernel_matrix", C.c_int, [_vp, _KD, _dp, _i64, _dp, _i64, _i32, _dp]),
    ("mln_kernel_grad", C.c_int, [_vp, _KD, _dp, _i64, _dp, _i64, _i32, _dp]),
    ("mln_predict_gradient", C.c_int, [_vp, _KD, _dp, _i64, _i32, _dp, _i64, _dp, _dp]),
    ("mln_predict_hessian", C.c_int, [_vp, _KD, _dp, _i64, _i32, _dp, _i64, _dp, _dp]),
    ("mln_kernel_gram", C.c_int, [_vp, _KD, _dp, _i64, _i32, _dp, _i64, _dp]),
    ("mln_nn_distances", C.c_int, [_vp, _dp, _i64, _dp, _i64, _i32, _i64, _dp]),
    ("mln_kmeans", C.c_int, [_vp, _dp, _i64, _i32, _i64, _i64, _i32, _dbl, _dp, C.POINTER(_i32), C.POINTER(_dbl)]),
    ("mln_kmeans_sklearn", C.c_int, [_vp, _dp, _i64, _i32, _i64, _i64, _dp, _i32, _i32, _dbl, _dp, _vp, C.POINTER(_i32),
                                     C.POINTER(_dbl)]),
    ("mln_chol_lower", C.c_int, [_vp, _dp, _i64, _dbl]),
    ("mln_trsm_lower", C.c_int, [_vp, _dp, _i64, _i32, _dp, _i64]),
    ("mln_fit_prepare", C.c_int, [_vp, _KD, _dp, _i64, _i32, _dp, _i64, _dbl, _dp, _i32, C.POINTER(_vp)]),
    ("mln_fit_from_L", C.c_int, [_vp, _dp, _i64, _i64, _dp, C.POINTER(_vp)]),
    ("mln_fit_prepare_from_K", C.c_int, [_vp, _dp, _i64, _i64, _dbl, _dp, _i32, C.POINTER(_vp)]),
    ("mln_fit_set_K_rows", C.c_int, [_vp, _i64, _i64, _dp]),
    ("mln_fit_finish_K", C.c_int, [_vp]),
    ("mln_gemm", C.c_int, [_vp, _i32, _i32, _i64, _i64, _i64, _dbl, _dp, _i64, _dp, _i64, _dbl, _dp, _i64]),
    ("mln_ewise", C.c_int, [_vp, _i32, _dp, _dp, _dbl, _dp, _i64]),
    ("mln_fit_destroy", None, [_vp]),
    ("mln_fit_get_Lp", C.c_int, [_vp, _dp]),
    ("mln_fit_get_L", C.c_int, [_vp, _i64, _i64, _dp]),
    ("mln_fit_rank", C.c_int, [_vp, C.POINTER(_i64)]),
    ("mln_eigh", C.c_int, [_vp, _dp, _i64, _dp, _dp, C.POINTER(_i32)]),
    ("mln_fit_gram_eigh", C.c_int, [_vp, _dp, C.POINTER(_i32)]),
    ("mln_fit_gram_rank", C.c_int, [_vp, _dbl, C.POINTER(_i64), C.POINTER(_dbl)]),
    ("mln_fit_project", C.c_int, [_vp, _i64, C.POINTER(_vp)]),
    ("mln_ridge_init", C.c_int, [_vp, _dp, _dp]),
    ("mln_precond_build", C.c_int, [_vp, _i64]),
    ("mln_fit_set_row_offset", C.c_int, [_vp, _i64]),
    ("mln_precond_apply", C.c_int, [_vp, _i32, _dp, _dp]),
    ("mln_objective_precond", C.c_int, [_vp, _dp, C.POINTER(_dbl), _dp, _dp]),
    ("mln_map_solve", C.c_int, [_vp, _dp, C.POINTER(SolverOpts), _dp, C.POINTER(_dbl), C.POINTER(_i32),
                                C.POINTER(_i32), C.POINTER(_i32)]),
    ("mln_fit_set_likelihood", C.c_int, [_vp, _dp, _dp, _dbl]),
    ("mln_objective", C.c_int, [_vp, _dp, C.POINTER(_dbl), _dp, _dp]),
    ("mln_transform", C.c_int, [_vp, _dp, _dbl, _dp]),
    ("mln_weights_cholesky", C.c_int, [_vp, _dp, _dp]),
    ("mln_weights_full", C.c_int, [_vp, _dp, _i64, _dbl, _dp]),
    ("mln_sparse_solve", C.c_int, [_vp, _KD, _dp, _i64, _i32, _dp, _i64, _dp, _i64, _dbl, _dbl, _dbl, _dp]),
    ("mln_sparse_solve_factors", C.c_int, [_vp, _KD, _dp, _i64, _i32, _dp, _i64, _dp, _i64, _dbl, _dbl, _dbl, _dp,
                                           _dp, _dp]),
    ("mln_sparse_solve_noise", C.c_int, [_vp, _KD, _dp, _i64, _i32, _dp, _i64, _dp, _i64, _dbl, _dp, _i32, _dbl, _dp]),
    ("mln_full_conditional_noise", C.c_int, [_vp, _KD, _dp, _i64, _i32, _dp, _i64, _dbl, _dp, _dbl, _dp, _dp, _dp, _dp]),
    ("mln_landmark_leverage", C.c_int, [_vp, _KD, _dp, _i64, _i32, _dp, _i64, _dp, _dp, _i64, _dbl, _dp]),
    ("mln_predict_mean", C.c_int, [_vp, _KD, _dp, _i64, _i32, _dp, _i64, _dp, _i64, _dbl, _dp]),
    ("mln_predict_covariance", C.c_int, [_vp, _KD, _dp, _i64, _i32, _dp, _i64, _dp, _i32, _dp]),
    ("mln_predict_mean_covariance", C.c_int, [_vp, _KD, _dp, _i64, _i32, _dp, _i64, _dp, _i64, _i32, _dp]),
    ("mln_stage_times", C.c_int, [_vp, _dp]),
    ("mln_diag_peak", C.c_int, [_vp, _i32, _i64, C.POINTER(_dbl)]),
    ("mln_diag_overlap", C.c_int, [_vp, _i64, _i64, _i32, _i64, _dp]),
    ("mln_diag_gram_i8", C.c_int, [_vp, _vp, _i64, _i64, _vp, _i32, C.POINTER(_dbl)]),
    ("mln_diag_dgemm", C.c_int, [_vp, _i32, _i32, _i64, _i64, _i64, _i32, _i32, _i32, C.POINTER(_dbl)]),
    ("mln_diag_dgemm_compare", C.c_int, [_vp, _i32, _i32, _i64, _i64, _i64, _i32, _i32, _dbl, _i32, C.POINTER(_dbl)]),
]

_lib = None
_lib_lock = threading.Lock()


def load_library():
    """dlopen libmellon_hip.so and declare every prototype.  Raises if it was not built."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise MellonHipError(
                f"{LIB_PATH} is missing: build it with `python -c \"import __graft_entry__ as g; g.build()\"` "
                "(hipcc --offload-arch=gfx950).  mellon_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)   # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def device_count():
    """GPUs visible to this process."""
    n = C.c_int(0)
    load_library().mln_device_count(C.byref(n))
    return int(n.value)


class BlockEvaluatedCov:
    """Marker base of covariances that do not lower to ONE device program (a user-defined Python `k`, or a tree beyond
    MLN_MAX_LEAVES / MLN_MAX_TOKS): `block(x, y)` returns the kernel values of one row block -- a DeviceArray when they
    were assembled on the device, else a host array -- and the Context methods below take the values-in route
    (mln_fit_prepare_from_K, mln_gemm).  base_cov.BlockCov implements it."""
    rows_per_block = 8192

    def block(self, x, y):
        raise NotImplementedError


def _needs_program(desc, what):
    if isinstance(desc, BlockEvaluatedCov):
        raise NotImplementedError(
            f"{what} needs a covariance that lowers to one device program (built-in kernels combined with + * ** within "
            "the program limits); user-defined Python kernels and larger trees support k(), fit and the predictive mean.")


class DeviceArray:
    """A float64 array resident in HBM (owned by a Context)."""

    def __init__(self, ctx, shape):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.size = int(np.prod(self.shape)) if self.shape else 1
        ptr = C.c_void_p()
        ctx._check(ctx.lib.mln_malloc(ctx.handle, self.size * 8, C.byref(ptr)))
        self.ptr = ptr.value

    @property
    def nbytes(self):
        return self.size * 8

    @property
    def ndim(self):
        return len(self.shape)

    def to_host(self):
        out = np.empty(self.shape, dtype=np.float64)
        self.ctx._check(self.ctx.lib.mln_memcpy(self.ctx.handle, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr is not None and self.ctx.handle is not None:
            self.ctx.lib.mln_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    """Pointer of a NumPy array or DeviceArray (None -> NULL)."""
    if a is None:
        return None
    if isinstance(a, DeviceArray):
        return a.ptr
    return a.ctypes.data


class Context:
    """One GPU, one HIP stream (mln_ctx).  Not thread-safe."""

    def __init__(self, device=None):
        self.lib = load_library()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) if "MELLON_AMD_DEVICE" not in os.environ \
                else int(os.environ["MELLON_AMD_DEVICE"])
        h = C.c_void_p()
        rc = self.lib.mln_ctx_create(int(device), C.byref(h))
        self.handle = None
        if rc != MLN_OK:
            msg = self.lib.mln_last_error(None).decode()
            raise MellonHipError(f"mln_ctx_create(device={device}) failed: {msg}")
        self.handle = h.value
        self.device = int(device)
        self.n_ranks, self.rank = 1, 0

    def close(self):
        if self.handle is not None:
            self.lib.mln_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, jitter=None):
        if rc == MLN_OK:
            return
        msg = self.lib.mln_last_error(self.handle).decode()
        if rc == MLN_ERR_NOT_PD:
            # same text as the reference (decomposition.py:116-122, conditional.py:74-80)
            raise ValueError(
                f"Covariance not positively definite with jitter={jitter}. "
                "Consider increasing the jitter for numerical stabilization.")
        if rc == MLN_ERR_SHAPE:
            raise ValueError(f"libmellon_hip: {msg}")
        if rc == MLN_ERR_UNSUPPORTED:
            raise NotImplementedError(f"libmellon_hip: {msg}")
        cause = None
        staged = getattr(self, "_host_collective", None)
        if staged is not None and getattr(staged[1], "failure", None) is not None:
            cause, staged[1].failure = staged[1].failure, None     # what the host-staged collective's callback ran into
        raise MellonHipError(f"libmellon_hip error {rc}: {msg}" + (f" ({type(cause).__name__}: {cause})" if cause else "")) from cause

    # -- info / memory ---------------------------------------------------------------------------
    def device_info(self):
        name = C.create_string_buffer(64)
        cu, mem = C.c_int(), C.c_int64()
        self._check(self.lib.mln_device_info(self.handle, name, 64, C.byref(cu), C.byref(mem)))
        return {"arch": name.value.decode(), "n_cu": cu.value, "mem_bytes": mem.value}

    def synchronize(self):
        self._check(self.lib.mln_synchronize(self.handle))

    def to_device(self, a):
        a = _f64(a)
        d = DeviceArray(self, a.shape)
        self._check(self.lib.mln_memcpy(self.handle, d.ptr, a.ctypes.data, a.nbytes))
        return d

    def empty(self, shape):
        return DeviceArray(self, shape)

    def pinned(self, a):
        """Context manager: the host array `a` page-locked for its duration (mln_host_register), so that fits which are
        handed `a` itself upload it by DMA under their kernels.  Registration costs ~0.1 ms per MB -- once per array."""
        return _Pinned(self, a)

    # -- communicator ------------------------------------------------------------------------------
    def comm_unique_id(self):
        buf = C.create_string_buffer(MLN_UNIQUE_ID_BYTES)
        self._check(self.lib.mln_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id, n_ranks, rank):
        buf = C.create_string_buffer(bytes(unique_id), MLN_UNIQUE_ID_BYTES)
        self._check(self.lib.mln_comm_init(self.handle, buf, int(n_ranks), int(rank)))
        self.n_ranks, self.rank = int(n_ranks), int(rank)

    def comm_init_host(self, n_ranks, rank, collective):
        """Device collectives staged through host memory: `collective(user, op, buf, buf2, count) -> 0 / 1` is called by
        the library with pinned host pointers (distributed.HostStagedCollectives)."""
        fn = HOST_COLLECTIVE_FN(collective)
        self._check(self.lib.mln_comm_init_host(self.handle, int(n_ranks), int(rank), C.cast(fn, C.c_void_p), None))
        self._host_collective = (fn, collective)          # keep the trampoline alive as long as the context
        self.n_ranks, self.rank = int(n_ranks), int(rank)

    def comm_info(self, timing=None, reset=False):
        """Transport, the rank count / rank the TRANSPORT reports, and the collectives carried so far (mln_comm_info)."""
        info = (_i32 * 4)()
        stats = np.zeros(10, dtype=np.float64)
        flags = (1 if timing is True else 2 if timing is False else 0) | (4 if reset else 0)
        self._check(self.lib.mln_comm_info(self.handle, info, stats.ctypes.data, flags))
        names = {0: "none", 1: "rccl", 2: "loopback", 3: "host"}
        return {"transport": names.get(int(info[0]), "?"), "ranks_reported_by_transport": int(info[1]),
                "rank_reported_by_transport": int(info[2]), "rccl_version_code": int(info[3]),
                "allreduce_calls": int(stats[0]), "allreduce_bytes": float(stats[1]),
                "broadcast_calls": int(stats[2]), "broadcast_bytes": float(stats[3]),
                "allgather_calls": int(stats[4]), "allgather_bytes": float(stats[5]),
                "small_allreduce_calls": int(stats[6]), "large_allreduce_ms": float(stats[7]),
                "broadcast_allgather_ms": float(stats[8]), "small_allreduce_ms": float(stats[9])}

    def comm_init_loopback(self, group, rank):
        """Join the in-process loopback communicator `group` (a LoopbackGroup) as `rank`."""
        self._check(self.lib.mln_comm_init_loopback(self.handle, group.handle, int(rank)))
        self._loop_group = group          # keeps the group alive as long as this context
        self.n_ranks, self.rank = group.n_ranks, int(rank)

    def allreduce_sum(self, a):
        if isinstance(a, DeviceArray):
            self._check(self.lib.mln_comm_allreduce_sum(self.handle, a.ptr, a.size))
            return a
        a = _f64(a).copy()
        self._check(self.lib.mln_comm_allreduce_sum(self.handle, a.ctypes.data, a.size))
        return a

    # -- operators -----------------------------------------------------------------------------------
    def kernel_matrix(self, desc, x, y):
        x, y = _as2d(x), _as2d(y)
        if x.shape[1] != y.shape[1]:
            raise ValueError("x and y must have the same number of features")
        if isinstance(desc, BlockEvaluatedCov):
            out = np.empty((x.shape[0], y.shape[0]), dtype=np.float64)
            for i0 in range(0, x.shape[0], desc.rows_per_block):
                blk = desc.block(x[i0:i0 + desc.rows_per_block], y)
                out[i0:i0 + desc.rows_per_block] = blk.to_host() if isinstance(blk, DeviceArray) else blk
            return out
        out = np.empty((x.shape[0], y.shape[0]), dtype=np.float64)
        self._check(self.lib.mln_kernel_matrix(self.handle, desc.ref, _ptr(x), x.shape[0], _ptr(y), y.shape[0],
                                               x.shape[1], out.ctypes.data))
        return out

    def kernel_gram(self, desc, x, xu):
        """cov(x, xu)^T cov(x, xu) as an (m, m) array."""
        if isinstance(desc, BlockEvaluatedCov):
            xh = x.to_host() if isinstance(x, DeviceArray) else _as2d(x)
            xu = _as2d(xu)
            G = self.to_device(np.zeros((xu.shape[0], xu.shape[0])))
            for i0 in range(0, xh.shape[0], desc.rows_per_block):
                blk = desc.block(xh[i0:i0 + desc.rows_per_block], xu)
                self.gemm(blk, blk, ta=True, out=G, beta=1.0)
            G = self.allreduce_sum(G)
            return G.to_host()
        x = x if isinstance(x, DeviceArray) else _as2d(x)
        xu = _as2d(xu)
        out = np.empty((xu.shape[0], xu.shape[0]), dtype=np.float64)
        self._check(self.lib.mln_kernel_gram(self.handle, desc.ref, _ptr(x), x.shape[0], x.shape[1], _ptr(xu),
                                             xu.shape[0], out.ctypes.data))
        return out

    def kernel_grad(self, desc, x, y):
        """d cov(x_i, y_j) / d y_j as an (n, m, d) array (Covariance.k_grad)."""
        _needs_program(desc, "kernel_grad")
        x, y = _as2d(x), _as2d(y)
        if x.shape[1] != y.shape[1]:
            raise ValueError("x and y must have the same number of features")
        out = np.empty((x.shape[0], y.shape[0], x.shape[1]), dtype=np.float64)
        self._check(self.lib.mln_kernel_grad(self.handle, desc.ref, _ptr(x), x.shape[0], _ptr(y), y.shape[0],
                                             x.shape[1], out.ctypes.data))
        return out

    def predict_gradient(self, desc, xnew, centers, W):
        """Gradient of the predictive mean with respect to each query point: (n_new, d)."""
        _needs_program(desc, "predict_gradient")
        xnew = xnew if isinstance(xnew, DeviceArray) else _as2d(xnew)
        centers = centers if isinstance(centers, DeviceArray) else _as2d(centers)
        Wd = _f64(W)
        if Wd.ndim != 1 or Wd.shape[0] != centers.shape[0]:
            raise NotImplementedError("gradients are available for single-output predictors (weights of shape (m,))")
        n_new, d = xnew.shape
        out = np.empty((n_new, d), dtype=np.float64)
        self._check(self.lib.mln_predict_gradient(self.handle, desc.ref, _ptr(xnew), n_new, d, _ptr(centers),
                                                  centers.shape[0], Wd.ctypes.data, out.ctypes.data))
        return out


    def predict_hessian(self, desc, xnew, centers, W):
        """Hessian of the predictive mean at each query point: (n_new, d, d)."""
        _needs_program(desc, "predict_hessian")
        xnew = xnew if isinstance(xnew, DeviceArray) else _as2d(xnew)
        centers = centers if isinstance(centers, DeviceArray) else _as2d(centers)
        Wd = _f64(W)
        if Wd.ndim != 1 or Wd.shape[0] != centers.shape[0]:
            raise NotImplementedError("hessians are available for single-output predictors (weights of shape (m,))")
        n_new, d = xnew.shape
        out = np.empty((n_new, d, d), dtype=np.float64)
        self._check(self.lib.mln_predict_hessian(self.handle, desc.ref, _ptr(xnew), n_new, d, _ptr(centers),
                                                 centers.shape[0], Wd.ctypes.data, out.ctypes.data))
        return out

    def nn_distances(self, x, y=None, self_offset=0):
        """Exact nearest-neighbour distance of each row of x among the rows of y (default: x itself)."""
        x = x if isinstance(x, DeviceArray) else _as2d(x)
        y = x if y is None else (y if isinstance(y, DeviceArray) else _as2d(y))
        out = np.empty(x.shape[0], dtype=np.float64)
        self._check(self.lib.mln_nn_distances(self.handle, _ptr(x), x.shape[0], _ptr(y), y.shape[0], x.shape[1],
                                              int(self_offset), out.ctypes.data))
        return out

    def kmeans(self, x, m, seed=42, max_iter=300, tol=1e-4, return_info=False, init="device"):
        """k-means++ / Lloyd centroids on the device (m x d).  init="device": the library's own draws (fastest);
        init="sklearn": the seeds sklearn.cluster.k_means(x, m, n_init=1, random_state=seed) picks -- numpy's
        RandomState(seed) supplies exactly the numbers sklearn's _kmeans_plusplus consumes (mln_kmeans_sklearn)."""
        x = x if isinstance(x, DeviceArray) else _as2d(x)
        centers = np.empty((int(m), x.shape[1]), dtype=np.float64)
        nit, inertia = C.c_int32(), C.c_double()
        if init == "sklearn":
            if not sklearn_seeding_compatible():
                import warnings
                warnings.warn("kmeans(init='sklearn'): the installed sklearn does not draw its k-means++ seeds the way this "
                              "library replays them (or is not importable); the seeds are k-means++ seeds of RandomState(seed), "
                              "but not necessarily the cells sklearn.cluster.k_means would pick.", RuntimeWarning)
            first, uni, trials = sklearn_seeding_numbers(x.shape[0], int(m), seed)
            idx = np.empty(int(m), dtype=np.int64)
            self._check(self.lib.mln_kmeans_sklearn(self.handle, _ptr(x), x.shape[0], x.shape[1], int(m), int(first),
                                                    uni.ctypes.data, trials, int(max_iter), float(tol), centers.ctypes.data,
                                                    idx.ctypes.data, C.byref(nit), C.byref(inertia) if return_info else None))
            self.last_kmeans_seed_indices = idx
            return (centers, nit.value, inertia.value) if return_info else centers
        if init != "device":
            raise ValueError("init must be 'device' or 'sklearn'")
        self._check(self.lib.mln_kmeans(self.handle, _ptr(x), x.shape[0], x.shape[1], int(m), int(seed),
                                        int(max_iter), float(tol), centers.ctypes.data, C.byref(nit),
                                        C.byref(inertia) if return_info else None))   # (the inertia is a full fp64 assignment)
        return (centers, nit.value, inertia.value) if return_info else centers

    def chol_lower(self, A, add_diag=0.0, jitter=None):
        A = _f64(A).copy()
        if A.ndim != 2 or A.shape[0] != A.shape[1]:
            raise ValueError("A must be square")
        self._check(self.lib.mln_chol_lower(self.handle, A.ctypes.data, A.shape[0], float(add_diag)),
                    jitter=add_diag if jitter is None else jitter)
        return A

    def eigh(self, A):
        """(w ascending, V with eigenvectors as columns) of the symmetric A -- jnp.linalg.eigh."""
        A = _f64(A)
        if A.ndim != 2 or A.shape[0] != A.shape[1]:
            raise ValueError("A must be square")
        m = A.shape[0]
        w, V, sw = np.empty(m), np.empty((m, m)), _i32(0)
        self._check(self.lib.mln_eigh(self.handle, A.ctypes.data, m, w.ctypes.data, V.ctypes.data, C.byref(sw)))
        self.last_eigh_sweeps = int(sw.value)
        return w, V

    def trsm_lower(self, Lf, B, trans=False):
        Lf = _f64(Lf)
        B2 = _f64(B).copy()
        shape = B2.shape
        B2 = B2.reshape(Lf.shape[0], -1)
        self._check(self.lib.mln_trsm_lower(self.handle, Lf.ctypes.data, Lf.shape[0], 1 if trans else 0,
                                            B2.ctypes.data, B2.shape[1]))
        return B2.reshape(shape)

    def predict_mean(self, desc, xnew, centers, W, mu, out=None):
        """mu + cov(xnew, centers) W.  out: a DeviceArray of the result's shape keeps the predictions in HBM (returned
        as is); default: a new host array."""
        xnew = xnew if isinstance(xnew, DeviceArray) else _as2d(xnew)
        centers = centers if isinstance(centers, DeviceArray) else _as2d(centers)
        Wd = W if isinstance(W, DeviceArray) else _f64(W)
        n_new, d = xnew.shape
        m = centers.shape[0]
        p = 1 if len(Wd.shape) == 1 else Wd.shape[1]
        shape = (n_new,) if len(Wd.shape) == 1 else (n_new, p)
        if out is not None:
            if not isinstance(out, DeviceArray) or tuple(out.shape) != shape:
                raise ValueError(f"out must be a DeviceArray of shape {shape}")
            _needs_program(desc, "predict_mean(out=DeviceArray)")
            self._check(self.lib.mln_predict_mean(self.handle, desc.ref, _ptr(xnew), n_new, d, _ptr(centers), m,
                                                  _ptr(Wd), p, float(mu), out.ptr))
            return out
        out = np.empty(shape, dtype=np.float64)
        if isinstance(desc, BlockEvaluatedCov):
            # mean = mu + cov(Xnew, centers) W with the kernel block from the user's k: conditional.py:366-373,651-658,899-906
            xh = xnew.to_host() if isinstance(xnew, DeviceArray) else xnew
            ch = centers.to_host() if isinstance(centers, DeviceArray) else centers
            W2 = Wd if isinstance(Wd, DeviceArray) else self.to_device(np.ascontiguousarray(Wd.reshape(m, p)))
            for i0 in range(0, n_new, desc.rows_per_block):
                blk = desc.block(xh[i0:i0 + desc.rows_per_block], ch)
                res = self.gemm(blk, W2).to_host() + float(mu)
                out[i0:i0 + desc.rows_per_block] = res[:, 0] if out.ndim == 1 else res
            return out
        self._check(self.lib.mln_predict_mean(self.handle, desc.ref, _ptr(xnew), n_new, d, _ptr(centers), m,
                                              _ptr(Wd), p, float(mu), out.ctypes.data))
        return out

    def _rows_sq_blocks(self, desc, xnew, centers, Mt, diag):
        """|cov(x_i, centers) Mt|^2 per row (diag) or the Gram of those rows: blocks of the user's kernel times a device
        matrix on the matrix cores."""
        xnew, centers = _as2d(xnew), _as2d(centers)
        Md = self.to_device(np.ascontiguousarray(Mt))
        n = xnew.shape[0]
        if not diag:
            blk = desc.block(xnew, centers)
            T = self.gemm(blk if isinstance(blk, DeviceArray) else self.to_device(blk), Md)
            return self.gemm(T, T, tb=True).to_host()
        out = np.empty(n)
        ones = self.to_device(np.ones((Md.shape[1], 1)))
        for i0 in range(0, n, desc.rows_per_block):
            blk = desc.block(xnew[i0:i0 + desc.rows_per_block], centers)
            T = self.gemm(blk if isinstance(blk, DeviceArray) else self.to_device(blk), Md)
            out[i0:i0 + desc.rows_per_block] = self.gemm(self.ewise(OP_MUL, T, T), ones).to_host()[:, 0]
        return out

    def predict_covariance(self, desc, xnew, centers, Lf, diag=True):
        if isinstance(desc, BlockEvaluatedCov):
            # k(x, x) - |cov(x, centers) L^-T|^2 (conditional.py:375-381,660-685,707-716)
            Lf = _f64(Lf)
            LinvT = np.ascontiguousarray(self.trsm_lower(Lf, np.eye(Lf.shape[0])).T)
            kss = desc.diag(_as2d(xnew)) if diag else desc.block(_as2d(xnew), _as2d(xnew))
            kss = kss.to_host() if isinstance(kss, DeviceArray) else kss
            return kss - self._rows_sq_blocks(desc, xnew, centers, LinvT, diag)
        xnew, centers, Lf = _as2d(xnew), _as2d(centers), _f64(Lf)
        n = xnew.shape[0]
        out = np.empty((n,) if diag else (n, n), dtype=np.float64)
        self._check(self.lib.mln_predict_covariance(self.handle, desc.ref, _ptr(xnew), n, xnew.shape[1],
                                                    _ptr(centers), centers.shape[0], _ptr(Lf), 1 if diag else 0,
                                                    out.ctypes.data))
        return out

    def predict_mean_covariance(self, desc, xnew, centers, W, diag=True):
        if isinstance(desc, BlockEvaluatedCov):
            return self._rows_sq_blocks(desc, xnew, centers, _as2d(W), diag)
        xnew, centers, W = _as2d(xnew), _as2d(centers), _as2d(W)
        n = xnew.shape[0]
        out = np.empty((n,) if diag else (n, n), dtype=np.float64)
        self._check(self.lib.mln_predict_mean_covariance(self.handle, desc.ref, _ptr(xnew), n, xnew.shape[1],
                                                         _ptr(centers), centers.shape[0], _ptr(W), W.shape[1],
                                                         1 if diag else 0, out.ctypes.data))
        return out

    def sparse_solve(self, desc, x, xu, y, mu, sigma, jitter, return_factors=False):
        """Weights of the noisy landmark conditional; with return_factors also (Lp, Cs = Lp L_B)."""
        if isinstance(desc, BlockEvaluatedCov):
            W, Lp, Cs = self._sparse_solve_blocks(desc, x, xu, y, mu, np.array([float(sigma)]), jitter, return_factors)
            return (W, Lp, Cs) if return_factors else W
        x = x if isinstance(x, DeviceArray) else _as2d(x)
        xu = _as2d(xu)
        y2 = y if isinstance(y, DeviceArray) else _f64(y)          # (HBM-resident targets are used in place)
        y_ndim = len(y2.shape)
        m = xu.shape[0]
        p = 1 if y_ndim == 1 else y2.shape[1]
        W = np.empty((m,) if y_ndim == 1 else (m, p), dtype=np.float64)
        if not return_factors:
            self._check(self.lib.mln_sparse_solve(self.handle, desc.ref, _ptr(x), x.shape[0], x.shape[1], _ptr(xu),
                                                  m, _ptr(y2), p, float(mu), float(sigma),
                                                  float(jitter), W.ctypes.data), jitter=jitter)
            return W
        Lp, Cs = np.empty((m, m)), np.empty((m, m))
        self._check(self.lib.mln_sparse_solve_factors(self.handle, desc.ref, _ptr(x), x.shape[0], x.shape[1],
                                                      _ptr(xu), m, _ptr(y2), p, float(mu), float(sigma),
                                                      float(jitter), W.ctypes.data, Lp.ctypes.data,
                                                      Cs.ctypes.data), jitter=jitter)
        return W, Lp, Cs

    SIGMA_SCALAR, SIGMA_PER_OUTPUT, SIGMA_PER_CELL = 0, 1, 2

    def _sparse_solve_blocks(self, desc, x, xu, y, mu, sigma, jitter, return_factors):
        """The landmark conditional (conditional.py:57-66,526-545) of a block-evaluated covariance, assembled from the
        library's device primitives: G = K^T K and R = K^T (y - mu) accumulate over row blocks on the matrix cores
        (mln_gemm) and are summed over the ranks; the m x m part -- Lp = chol(K_uu + jitter I), A A^T = Lp^-1 G Lp^-T,
        L_B = chol(A A^T / sigma^2 + I), the two pairs of triangular solves -- goes through mln_chol_lower /
        mln_trsm_lower.  sigma: one value, or one per output column (columns that share a value share the factor)."""
        xh = x.to_host() if isinstance(x, DeviceArray) else _as2d(x)
        xu = _as2d(xu)
        yh = y.to_host() if isinstance(y, DeviceArray) else _f64(y)
        y_ndim = yh.ndim
        r = (yh - float(mu)).reshape(xh.shape[0], -1)
        m, p = xu.shape[0], r.shape[1]
        if sigma.shape[0] not in (1, p):
            raise ValueError(f"sigma has shape {sigma.shape}, expected (1,) or ({p},)")
        G, R = self.to_device(np.zeros((m, m))), self.to_device(np.zeros((m, p)))
        for i0 in range(0, xh.shape[0], desc.rows_per_block):
            blk = desc.block(xh[i0:i0 + desc.rows_per_block], xu)
            blk = blk if isinstance(blk, DeviceArray) else self.to_device(blk)
            self.gemm(blk, blk, ta=True, out=G, beta=1.0)
            self.gemm(blk, np.ascontiguousarray(r[i0:i0 + desc.rows_per_block]), ta=True, out=R, beta=1.0)
        G, R = self.allreduce_sum(G).to_host(), self.allreduce_sum(R).to_host()
        Kuu = desc.block(xu, xu)
        Kuu = Kuu.to_host() if isinstance(Kuu, DeviceArray) else Kuu
        Lp = self.chol_lower(Kuu, add_diag=float(jitter))
        T = self.trsm_lower(Lp, G)                                             # Lp^-1 G
        AAt = self.trsm_lower(Lp, np.ascontiguousarray(T.T))                   # Lp^-1 G^T Lp^-T (symmetric)
        AAt = 0.5 * (AAt + AAt.T)
        Ar = self.trsm_lower(Lp, R)                                            # A (y - mu)
        W = np.empty((m, p))
        Cs = None
        sig_cols = np.broadcast_to(sigma, (p,))
        for sg in np.unique(sig_cols):
            cols = np.nonzero(sig_cols == sg)[0]
            s2 = float(sg) ** 2
            L_B = self.chol_lower(AAt / s2, add_diag=1.0)
            c = self.trsm_lower(L_B, np.ascontiguousarray(Ar[:, cols]) / s2)
            W[:, cols] = self.trsm_lower(Lp, self.trsm_lower(L_B, c, trans=True), trans=True)
            if return_factors:
                Cs = self.gemm(Lp, L_B).to_host()
        W = W[:, 0] if y_ndim == 1 else W
        return W, Lp, Cs

    def sparse_solve_noise(self, desc, x, xu, y, mu, sigma, kind, jitter):
        """Landmark-conditional weights under per-output (sigma[p]) or per-cell (sigma[n]) noise."""
        if isinstance(desc, BlockEvaluatedCov):
            if kind == self.SIGMA_PER_CELL:
                _needs_program(desc, "sparse_solve_noise with one sigma per cell")
            return self._sparse_solve_blocks(desc, x, xu, y, mu, _f64(np.atleast_1d(sigma)), jitter, False)[0]
        x = x if isinstance(x, DeviceArray) else _as2d(x)
        xu = _as2d(xu)
        y2 = _f64(y)
        sig = _f64(np.atleast_1d(sigma))
        m = xu.shape[0]
        p = 1 if y2.ndim == 1 else y2.shape[1]
        want = {self.SIGMA_SCALAR: 1, self.SIGMA_PER_OUTPUT: p, self.SIGMA_PER_CELL: x.shape[0]}[kind]
        if sig.shape != (want,):
            raise ValueError(f"sigma has shape {sig.shape}, expected {(want,)} for noise kind {kind}")
        W = np.empty((m,) if y2.ndim == 1 else (m, p), dtype=np.float64)
        self._check(self.lib.mln_sparse_solve_noise(self.handle, desc.ref, _ptr(x), x.shape[0], x.shape[1], _ptr(xu),
                                                    m, y2.ctypes.data, p, float(mu), sig.ctypes.data, int(kind),
                                                    float(jitter), W.ctypes.data), jitter=jitter)
        return W

    def full_conditional_noise(self, desc, x, y, mu, sigma, jitter, leverage=False, obs_variance=False):
        """Full-GP weights for outputs with one noise level each, from one eigendecomposition of K(x, x).
        Returns W, or (W, leverage), or (W, leverage, corrected_r2, variance_W)."""
        _needs_program(desc, "full_conditional_noise")
        x = _as2d(x)
        y2 = np.ascontiguousarray(_f64(y))
        sig = _f64(np.atleast_1d(sigma))
        n, p = y2.shape
        if sig.shape != (p,) or x.shape[0] != n:
            raise ValueError("sigma must hold one value per output column and x, y must agree in length")
        leverage = leverage or obs_variance
        W = np.empty((n, p))
        H = np.empty((n, p)) if leverage else None
        Cr, VW = (np.empty((n, p)), np.empty((n, p))) if obs_variance else (None, None)
        ptr = lambda a: None if a is None else a.ctypes.data
        self._check(self.lib.mln_full_conditional_noise(self.handle, desc.ref, _ptr(x), n, x.shape[1], y2.ctypes.data, p,
                                                        float(mu), sig.ctypes.data, float(jitter), W.ctypes.data,
                                                        ptr(H), ptr(Cr), ptr(VW)), jitter=jitter)
        if obs_variance:
            return W, H, Cr, VW
        return (W, H) if leverage else W

    def pairwise_distance(self, x, y):
        """util.distance (util.py:351-366) through the kernel-matrix pass with the value-only DISTANCE leaf."""
        from .base_cov import LoweredCov
        d = _as2d(x).shape[1]
        desc = LoweredCov([(K_DISTANCE, 1.0, 1.0, np.arange(d))], [(OP_LEAF, 0, 0.0)])
        return self.kernel_matrix(desc, x, y)

    def landmark_leverage(self, desc, x, xu, Lk, sigma, jitter):
        """(n, p) leverage of the landmark conditional for the p noise levels `sigma`, K_uu = Lk Lk^T."""
        _needs_program(desc, "landmark_leverage")
        x = x if isinstance(x, DeviceArray) else _as2d(x)
        xu = _as2d(xu)
        Lk = _f64(Lk)
        sig = _f64(np.atleast_1d(sigma))
        m = xu.shape[0]
        if Lk.shape != (m, m):
            raise ValueError(f"Lk has shape {Lk.shape}, expected {(m, m)}")
        out = np.empty((x.shape[0], sig.shape[0]), dtype=np.float64)
        self._check(self.lib.mln_landmark_leverage(self.handle, desc.ref, _ptr(x), x.shape[0], x.shape[1], _ptr(xu), m,
                                                   Lk.ctypes.data, sig.ctypes.data, sig.shape[0], float(jitter),
                                                   out.ctypes.data), jitter=jitter)
        return out

    def diag_overlap(self, n, m, d, gram_rows):
        out = np.zeros(6)
        self._check(self.lib.mln_diag_overlap(self.handle, int(n), int(m), int(d), int(gram_rows), out.ctypes.data))
        return dict(zip(["k_ms", "gram_ms", "k_par_gram_ms", "chol_ms", "k_par_chol_ms", "k_par_chol_gram_ms"], out))

    def diag_gram_i8(self, a, reps=1):
        """(Gram of round(a * 8355711) / 8355711**2, ms per call): the preconditioner's integer Gram in isolation."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        out = np.empty((a.shape[1], a.shape[1]))
        ms = C.c_double()
        self._check(self.lib.mln_diag_gram_i8(self.handle, a.ctypes.data, a.shape[0], a.shape[1], out.ctypes.data, int(reps),
                                              C.byref(ms)))
        return out, ms.value

    def diag_peak(self, what, nbytes=1 << 32):
        r = C.c_double()
        self._check(self.lib.mln_diag_peak(self.handle, int(what), int(nbytes), C.byref(r)))
        return r.value

    def diag_dgemm(self, ta, tb, M, N, K, lower_only=0, split_k=1, reps=3):
        r = C.c_double()
        self._check(self.lib.mln_diag_dgemm(self.handle, ta, tb, M, N, K, lower_only, split_k, reps, C.byref(r)))
        return r.value

    def fit_prepare(self, desc, x, landmarks, jitter, Lp=None, implicit=False):
        if isinstance(desc, BlockEvaluatedCov):
            return Fit.from_blocks(self, desc, x, landmarks, jitter, Lp, implicit)
        return Fit(self, desc, x, landmarks, jitter, Lp, implicit)

    def diag_dgemm_compare(self, ta, tb, M, N, K, lower_only=0, kmode=0, beta=0.0, any_size=True):
        """(largest |mixed-tile kernel - single-size kernels|, largest |value|) on the same operands (mln_diag_dgemm_compare)."""
        r = (C.c_double * 2)()
        self._check(self.lib.mln_diag_dgemm_compare(self.handle, int(ta), int(tb), M, N, K, lower_only, kmode, float(beta),
                                                    1 if any_size else 0, r))
        return float(r[0]), float(r[1])

    def diag_dgemm_batch(self, ta, tb, M, N, K, lower_only=0, kmode=0, beta=0.0):
        """largest |one launch with batch = 2 - the two single launches| (mln_diag_dgemm_compare, any_size = 2)."""
        r = (C.c_double * 2)()
        self._check(self.lib.mln_diag_dgemm_compare(self.handle, int(ta), int(tb), M, N, K, lower_only, kmode, float(beta), 2, r))
        return float(r[0])

    def gemm(self, A, B, ta=False, tb=False, alpha=1.0, beta=0.0, out=None):
        """out = alpha op(A) op(B) + beta out on the fp64 matrix cores (mln_gemm); A, B, out host arrays or
        DeviceArrays (2-D, row-major).  Returns a DeviceArray unless `out` is a host array."""
        A = A if isinstance(A, DeviceArray) else _as2d(A)
        B = B if isinstance(B, DeviceArray) else _as2d(B)
        M, K = (A.shape[1], A.shape[0]) if ta else A.shape
        Kb, N = (B.shape[1], B.shape[0]) if tb else B.shape
        if K != Kb:
            raise ValueError(f"gemm: inner dimensions differ ({K} vs {Kb})")
        if out is None:
            out = self.empty((M, N))
            beta = 0.0
        if tuple(out.shape) != (M, N):
            raise ValueError(f"gemm: out has shape {tuple(out.shape)}, expected {(M, N)}")
        self._check(self.lib.mln_gemm(self.handle, 1 if ta else 0, 1 if tb else 0, M, N, K, float(alpha), _ptr(A),
                                      A.shape[1], _ptr(B), B.shape[1], float(beta), _ptr(out), N))
        return out

    def ewise(self, op, a, b, out=None):
        """a op b element-wise on the device (op: OP_ADD / OP_MUL / OP_POW; b an array like a, or a scalar)."""
        a = a if isinstance(a, DeviceArray) else self.to_device(a)
        out = a if out is None else out
        scalar = 0.0
        bp = None
        if isinstance(b, (int, float, np.floating, np.integer)):
            scalar = float(b)
        else:
            b = b if isinstance(b, DeviceArray) else self.to_device(b)
            if tuple(b.shape) != tuple(a.shape):
                raise ValueError("ewise: shapes differ")
            bp = b.ptr
        self._check(self.lib.mln_ewise(self.handle, int(op), a.ptr, bp, scalar, out.ptr, a.size))
        return out


class LoopbackGroup:
    """mln_loopback: the device side of n thread-ranks of one process (distributed.run_loopback)."""

    def __init__(self, n_ranks):
        self.lib = load_library()
        h = C.c_void_p()
        if self.lib.mln_loopback_create(int(n_ranks), C.byref(h)) != MLN_OK:
            raise MellonHipError(f"mln_loopback_create({n_ranks}) failed")
        self.handle, self.n_ranks = h.value, int(n_ranks)

    def abort(self):
        """Release every rank waiting in a collective with an error (a peer failed elsewhere)."""
        if self.handle is not None:
            self.lib.mln_loopback_abort(self.handle)

    def __del__(self):
        try:
            if self.handle is not None:
                self.lib.mln_loopback_destroy(self.handle)
            self.handle = None
        except Exception:
            pass


def _as2d(a):
    a = _f64(a)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    return np.ascontiguousarray(a)


def sklearn_seeding_numbers(n, m, seed):
    """(first cell, (m - 1) x L uniform numbers, L) as sklearn's _kmeans_plusplus draws them from RandomState(seed):
    `random_state.choice(n, p=sample_weight / sample_weight.sum())` with unit weights, then per centre
    `random_state.uniform(size=L)`, L = 2 + int(log m)."""
    rs = np.random.RandomState(seed)
    trials = 2 + int(np.log(m))
    sw = np.ones(n, dtype=np.float64)
    first = int(rs.choice(n, p=sw / sw.sum()))
    uni = np.ascontiguousarray(rs.uniform(size=(max(m - 1, 0), trials)), dtype=np.float64)
    return first, uni, trials


_sklearn_seeding_ok = None


def sklearn_seeding_compatible():
    """Does the installed sklearn draw its k-means++ seeds the way sklearn_seeding_numbers() assumes (first centre by
    `RandomState.choice(n, p=...)`, 2 + int(log m) uniform trials per further centre)?  Checked ONCE against
    sklearn.cluster.kmeans_plusplus on 60 well-separated cells: all seed indices must agree with a NumPy replay of the
    published algorithm fed by sklearn_seeding_numbers.  False (with a warning from the caller) when sklearn is another
    version or not importable -- the seeds are then valid k-means++ seeds, but not sklearn's.
    Tie rule of the device search (documented difference): a cumulative-potential target selects the first cell with
    excl <= target < incl and cells at distance zero are never selected, where NumPy's searchsorted(side="left") takes the
    first cell with incl >= target; the two differ only on exact ties of the cumulative sum or a target of exactly 0."""
    global _sklearn_seeding_ok
    if _sklearn_seeding_ok is not None:
        return _sklearn_seeding_ok
    try:
        from sklearn.cluster import kmeans_plusplus
        n, m, seed = 60, 6, 11
        g = np.random.RandomState(5)
        x = np.ascontiguousarray(g.normal(size=(n, 3)) + 20.0 * g.randint(0, 6, size=(n, 1)))
        _, want = kmeans_plusplus(x, m, random_state=seed)
        first, uni, trials = sklearn_seeding_numbers(n, m, seed)
        got = [first]
        d2 = ((x - x[first]) ** 2).sum(1)
        for c in range(1, m):
            cand = np.searchsorted(np.cumsum(d2), uni[c - 1] * d2.sum())
            cand = np.clip(cand, None, n - 1)
            dc = ((x[cand][:, None, :] - x[None, :, :]) ** 2).sum(2)
            pots = np.minimum(d2[None, :], dc).sum(1)
            best = int(np.argmin(pots))
            got.append(int(cand[best]))
            d2 = np.minimum(d2, dc[best])
        _sklearn_seeding_ok = [int(v) for v in want] == got
    except Exception:
        _sklearn_seeding_ok = False
    return _sklearn_seeding_ok


class _Pinned:
    def __init__(self, ctx, a):
        if not (isinstance(a, np.ndarray) and a.flags.c_contiguous):
            raise TypeError("pinned(): a C-contiguous NumPy array is required")
        self.ctx, self.a, self.on = ctx, a, False

    def __enter__(self):
        self.ctx._check(self.ctx.lib.mln_host_register(self.ctx.handle, self.a.ctypes.data, self.a.nbytes))
        self.on = True
        return self.a

    def __exit__(self, *exc):
        if self.on:
            self.on = False
            self.ctx._check(self.ctx.lib.mln_host_unregister(self.ctx.handle, self.a.ctypes.data))
        return False


class Fit:
    """Device-resident shard state of one estimator fit (mln_fit)."""

    @classmethod
    def from_L(cls, ctx, L, Lp=None):
        """Adopt a precomputed factor (the estimator's `L=` ctor argument)."""
        self = cls.__new__(cls)
        self.ctx, self.lib, self.handle = ctx, ctx.lib, None
        L = L if isinstance(L, DeviceArray) else _as2d(L)
        n, m = L.shape
        Lp_ = None if Lp is None else _f64(Lp)
        if Lp_ is not None and Lp_.shape != (m, m):
            raise ValueError(f"Lp has shape {Lp_.shape}, expected {(m, m)}")
        h = C.c_void_p()
        ctx._check(self.lib.mln_fit_from_L(ctx.handle, _ptr(L), n, m, _ptr(Lp_), C.byref(h)))
        self.handle = h.value
        self.n, self.d, self.m, self.jitter = n, None, m, None
        self.implicit = False
        self._has_lp = Lp_ is not None
        return self

    @classmethod
    def from_blocks(cls, ctx, cov, x, landmarks, jitter, Lp=None, implicit=False):
        """The fit of a covariance that is evaluated block-wise by the binding (BlockEvaluatedCov: a user-defined Python
        `k`, or a tree too large for one device program): cov(xu, xu) and the row blocks of cov(x, xu) are handed to the
        library as VALUES (mln_fit_prepare_from_K / mln_fit_set_K_rows / mln_fit_finish_K); the factorisations, the Ridge
        start, the MAP solve and the weights are the unchanged device path."""
        self = cls.__new__(cls)
        self.ctx, self.lib, self.handle = ctx, ctx.lib, None
        xh = x.to_host() if isinstance(x, DeviceArray) else _as2d(x)
        n, d = xh.shape
        full = landmarks is None
        xu = xh if full else (landmarks.to_host() if isinstance(landmarks, DeviceArray) else _as2d(landmarks))
        m = xu.shape[0]
        self.implicit = bool(implicit) and not full
        Lp_ = None if Lp is None else _f64(Lp)
        if Lp_ is not None and Lp_.shape != (m, m):
            raise ValueError(f"Lp has shape {Lp_.shape}, expected {(m, m)}")
        Kuu = None
        if Lp_ is None:
            Kuu = cov.block(xu, xu)
        flags = (1 if self.implicit else 0) | (2 if full else 0)
        h = C.c_void_p()
        ctx._check(self.lib.mln_fit_prepare_from_K(ctx.handle, _ptr(Kuu), n, m, float(jitter), _ptr(Lp_), flags,
                                                   C.byref(h)), jitter=jitter)
        self.handle = h.value
        self.n, self.d, self.m, self.jitter = n, d, m, jitter
        self._has_lp = True
        if not full:
            for i0 in range(0, n, cov.rows_per_block):
                blk = cov.block(xh[i0:i0 + cov.rows_per_block], xu)
                rows = min(cov.rows_per_block, n - i0)
                if tuple(blk.shape) != (rows, m):
                    raise ValueError(f"covariance returned a block of shape {tuple(blk.shape)}, expected {(rows, m)}")
                blk = blk if isinstance(blk, DeviceArray) else _f64(blk)
                ctx._check(self.lib.mln_fit_set_K_rows(self.handle, i0, rows, _ptr(blk)))
        ctx._check(self.lib.mln_fit_finish_K(self.handle))
        return self

    def gram_eigh(self):
        """Eigenvalues (ascending) of L^T L over all cells of all ranks; eigenvectors stay on the device."""
        w, sw = np.empty(self.m), _i32(0)
        self._check(self.lib.mln_fit_gram_eigh(self.handle, w.ctypes.data, C.byref(sw)))
        self.last_eigh_sweeps = int(sw.value)
        return w

    def gram_rank(self, tol):
        """(number of singular values of L above tol * the largest, the largest): Sturm counts on the tridiagonalised
        Gram of all cells of all ranks -- no eigendecomposition."""
        r, smax = _i64(0), C.c_double()
        self._check(self.lib.mln_fit_gram_rank(self.handle, float(tol), C.byref(r), C.byref(smax)))
        return int(r.value), smax.value

    def project(self, p):
        """New handle with L <- L U[:, -p:] (top-p eigenvectors of L^T L, after gram_eigh)."""
        new = Fit.__new__(Fit)
        new.ctx, new.lib, new.handle = self.ctx, self.lib, None
        h = C.c_void_p()
        self._check(self.lib.mln_fit_project(self.handle, int(p), C.byref(h)))
        new.handle = h.value
        new.n, new.d, new.m, new.jitter = self.n, None, int(p), None
        new.implicit = False
        new._has_lp = False
        return new

    def __init__(self, ctx, desc, x, landmarks, jitter, Lp=None, implicit=False):
        self.ctx, self.lib, self.handle = ctx, ctx.lib, None
        self.implicit = bool(implicit) and landmarks is not None
        x = x if isinstance(x, DeviceArray) else _as2d(x)
        n, d = x.shape
        xu = None if landmarks is None else (landmarks if isinstance(landmarks, DeviceArray) else _as2d(landmarks))
        m = n if xu is None else xu.shape[0]
        Lp_ = None if Lp is None else _f64(Lp)
        if Lp_ is not None and Lp_.shape != (m, m):
            raise ValueError(f"Lp has shape {Lp_.shape}, expected {(m, m)}")
        h = C.c_void_p()
        # implicit fits: the landmark factor Lp is deferred (MLN_FIT_DEFER_LP = 4) -- it is factored in one chain of launches
        # with the preconditioner's matrix (mln_precond_build / mln_ridge_init), or by the first call that needs it
        flags = (1 | (4 if Lp_ is None else 0)) if self.implicit else 0
        ctx._check(self.lib.mln_fit_prepare(ctx.handle, desc.ref, _ptr(x), n, d, _ptr(xu), m, float(jitter),
                                            _ptr(Lp_), flags, C.byref(h)), jitter=jitter)
        self.handle = h.value
        self.n, self.d, self.m, self.jitter = n, d, m, jitter
        self._has_lp = True

    def _check(self, rc, jitter=None):
        """Context._check, with the reference's message for a deferred landmark factor that turns out not to be
        positive definite (decomposition.py:116-122: the jitter named is the fit's)."""
        if rc == MLN_ERR_NOT_PD and self.lib.mln_last_error(self.ctx.handle).startswith(b"cov(xu, xu)"):
            jitter = self.jitter
        self.ctx._check(rc, jitter=jitter)

    def close(self):
        if self.handle is not None and self.ctx.handle is not None:
            self.lib.mln_fit_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def Lp(self):
        out = np.empty((self.m, self.m), dtype=np.float64)
        self._check(self.lib.mln_fit_get_Lp(self.handle, out.ctypes.data))
        return out

    def L(self, row0=0, n_rows=None):
        n_rows = self.n - row0 if n_rows is None else n_rows
        out = np.empty((n_rows, self.m), dtype=np.float64)
        self._check(self.lib.mln_fit_get_L(self.handle, row0, n_rows, out.ctypes.data))
        return out

    def ridge_init(self, target):
        target = target if isinstance(target, DeviceArray) else _f64(target)
        z0 = np.empty(self.m, dtype=np.float64)
        self._check(self.lib.mln_ridge_init(self.handle, _ptr(target), z0.ctypes.data), jitter="ridge")
        return z0

    def set_likelihood(self, V, Vdr, mu):
        V = V if isinstance(V, DeviceArray) else _f64(V)
        Vdr = Vdr if isinstance(Vdr, DeviceArray) else _f64(Vdr)
        self._check(self.lib.mln_fit_set_likelihood(self.handle, _ptr(V), _ptr(Vdr), float(mu)))

    def objective(self, z, with_hess=False):
        z = _f64(z)
        loss = C.c_double()
        grad = np.empty(self.m, dtype=np.float64)
        hess = np.empty(self.m, dtype=np.float64) if with_hess else None
        self._check(self.lib.mln_objective(self.handle, z.ctypes.data, C.byref(loss), grad.ctypes.data,
                                               _ptr(hess)))
        return (loss.value, grad, hess) if with_hess else (loss.value, grad)

    def precond_build(self, row_stride=1, row_offset=0, force=False):
        """Factor the Ridge / preconditioner matrix from the cells whose global index (row_offset + local index) is a
        multiple of row_stride.  A no-op once a factor exists, unless `force` asks for this very sample (the library
        then re-factors when the stride differs from the one it has)."""
        built = getattr(self, "_precond_stride", None)
        if built is not None and (not force or built == int(row_stride)):
            return
        self._check(self.lib.mln_fit_set_row_offset(self.handle, int(row_offset)))
        self._check(self.lib.mln_precond_build(self.handle, int(row_stride)), jitter="ridge")
        self._precond_stride = max(1, int(row_stride))

    def precond_apply(self, mode, v):
        """mode 0: u = C^T z; 1: z = C^-T u; 2: g_u = C^-1 g_z  (C C^T = L^T L + I)."""
        v = _f64(v)
        out = np.empty(self.m, dtype=np.float64)
        self._check(self.lib.mln_precond_apply(self.handle, int(mode), v.ctypes.data, out.ctypes.data),
                        jitter="ridge")
        return out

    def objective_precond(self, u):
        """(loss, grad_u, z) of the preconditioned variable z = C^-T u."""
        u = _f64(u)
        loss = C.c_double()
        grad = np.empty(self.m, dtype=np.float64)
        z = np.empty(self.m, dtype=np.float64)
        self._check(self.lib.mln_objective_precond(self.handle, u.ctypes.data, C.byref(loss), grad.ctypes.data,
                                                       z.ctypes.data), jitter="ridge")
        return loss.value, grad, z

    def map_solve(self, z0, maxiter=5000, maxcor=10, maxls=30, ftol=1e-13, gtol=1e-7):
        """In-library L-BFGS on the preconditioned variable; returns (z, loss, n_eval, n_iter, status)."""
        z0 = _f64(z0)
        opts = SolverOpts(int(maxiter), int(maxcor), int(maxls), float(ftol), float(gtol))
        z = np.empty(self.m, dtype=np.float64)
        loss, nev, nit, st = C.c_double(), C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.lib.mln_map_solve(self.handle, z0.ctypes.data, C.byref(opts), z.ctypes.data,
                                               C.byref(loss), C.byref(nev), C.byref(nit), C.byref(st)),
                        jitter="ridge")
        return z, loss.value, nev.value, nit.value, st.value

    def transform(self, z, mu, out=None):
        z = _f64(z)
        ret = np.empty(self.n, dtype=np.float64) if out is None else out
        self._check(self.lib.mln_transform(self.handle, z.ctypes.data, float(mu), _ptr(ret)))
        return ret

    def weights_cholesky(self, z):
        z = _f64(z)
        w = np.empty(self.m, dtype=np.float64)
        self._check(self.lib.mln_weights_cholesky(self.handle, z.ctypes.data, w.ctypes.data))
        return w

    def weights_full(self, y, mu):
        y2 = _f64(y)
        p = 1 if y2.ndim == 1 else y2.shape[1]
        w = np.empty_like(y2)
        self._check(self.lib.mln_weights_full(self.handle, y2.ctypes.data, p, float(mu), w.ctypes.data))
        return w

    def stage_times(self):
        out = np.zeros(MLN_N_STAGE_TIMES, dtype=np.float64)
        self._check(self.lib.mln_stage_times(self.handle, out.ctypes.data))
        keys = ["kernel_matrix_s", "cholesky_s", "trsm_s", "ridge_gram_s", "ridge_solve_s",
                "objective_kernel_s", "objective_launches", "objective_bytes_per_launch",
                "objective32_kernel_s", "objective32_launches", "copy32_format", "emulation_excluded_s",
                "objective_sub_kernel_s", "objective_sub_launches", "objective_sub_stride", "precond_rebuild_s",
                "precond_rebuilds", "objective_pass_equivalents", "precond_rebuilds_declined", "precond_reverts",
                "start_halvings", "rank_path"]
        return dict(zip(keys, out.tolist()))


def release_cached_memory():
    """Return the library's cached device blocks to the driver."""
    load_library().mln_release_cached_memory()


_default_ctx = None


def default_context():
    """The context of the calling thread's communicator (thread-ranks, distributed.run_loopback), else the
    process-wide context on GPU ``LOCAL_RANK`` (or ``MELLON_AMD_DEVICE``)."""
    global _default_ctx
    from . import distributed
    comm = distributed.thread_state()
    if comm is not None and comm.ctx is not None:
        return comm.ctx
    if _default_ctx is None or _default_ctx.handle is None:
        _default_ctx = Context()
    return _default_ctx
