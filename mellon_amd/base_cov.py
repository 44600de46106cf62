"""Kernel-plugin surface: Covariance ABC with `+ * **`, active_dims and the reference's JSON
wire format (mellon/base_cov.py:17-497).  `k()` is evaluated by the HIP tile kernel through the
C-ABI: a covariance tree is lowered to a postfix program over leaves (include/mellon_hip.h).
A subclass that defines its own Python `k(x, y)` (the ABC's only contract, reference base_cov.py:17-69) cannot be
lowered: `lower()` then returns a `BlockCov`, through which the binding evaluates the USER'S function in row blocks and
hands the values to the library (mln_fit_prepare_from_K, mln_gemm) -- everything after the kernel matrix stays on the
device.  Trees larger than one device program (MLN_MAX_LEAVES / MLN_MAX_TOKS / stack depth) take the same route, their
sub-trees evaluated by device programs and combined on the device (mln_ewise).
"""
import ctypes as C
import json
import sys
from abc import ABC
from datetime import datetime
from importlib import import_module

import numpy as np

from . import _lib
from .util import compose_active_dims, deserialize, ensure_2d, make_serializable

MELLON_NAME = "mellon"       # class paths in serialized state follow the reference's package


class LoweredCov:
    """Owns the ctypes arrays behind one mln_kernel_desc."""

    def __init__(self, leaves, toks):
        self._dims = [np.ascontiguousarray(l[3], dtype=np.int32) for l in leaves]
        self._leaves = (_lib.Leaf * len(leaves))()
        for i, (kind, ls, alpha, dims) in enumerate(leaves):
            self._leaves[i].kind = kind
            self._leaves[i].ndims = len(self._dims[i])
            self._leaves[i].ls = float(ls)
            self._leaves[i].alpha = float(alpha)
            self._leaves[i].dims = self._dims[i].ctypes.data_as(C.POINTER(C.c_int32))
        self._toks = (_lib.Tok * len(toks))()
        for i, (op, leaf, val) in enumerate(toks):
            self._toks[i].op, self._toks[i].leaf, self._toks[i].value = op, leaf, float(val)
        self.desc = _lib.KernelDesc(len(leaves), len(toks), self._leaves, self._toks)
        self.ref = C.byref(self.desc)
        self.leaves, self.toks = leaves, toks


# limits of ONE device program (include/mellon_hip.h, csrc/api.hip mln_lower_cov)
MAX_LEAVES, MAX_TOKS, MAX_DEPTH, MAX_DIMS = 4, 16, 3, 256


def _program_fits(leaves, toks):
    if not (1 <= len(leaves) <= MAX_LEAVES and 1 <= len(toks) <= MAX_TOKS):
        return False
    if sum(len(l[3]) for l in leaves) > MAX_DIMS:
        return False
    depth = 0
    for op, _, _ in toks:
        depth += 1 if op in (_lib.OP_LEAF, _lib.OP_CONST) else -1
        if depth > MAX_DEPTH:
            return False
    return True


class BlockCov(_lib.BlockEvaluatedCov):
    """A covariance evaluated block by block: what `Covariance.lower()` returns for trees that are not one device
    program.  `block(x, y)` walks the tree: every sub-tree that IS a device program becomes one kernel-matrix launch,
    user-defined leaves call the user's `k` on the host, and Add / Mul / Pow nodes combine the blocks on the device."""

    def __init__(self, cov, d):
        self.cov, self.d = cov, int(d)

    def block(self, x, y):
        x = np.ascontiguousarray(ensure_2d(x), dtype=np.float64)
        y = np.ascontiguousarray(ensure_2d(y), dtype=np.float64)
        return self.cov._block(np.arange(self.d), x, y)

    def diag(self, x):
        """k(x_i, x_i) of every row (reference base_cov.py:71-93)."""
        return self.cov.diag(x)


class Covariance(ABC):
    """Base covariance function (reference base_cov.py:17-224)."""

    _kind = None  # built-in leaves set this to an mln_kind

    def __init__(self, active_dims=None):
        self.active_dims = active_dims

    def __str__(self):
        return self.__repr__()

    def __repr__(self):
        args = [f"{k}={v}" for k, v in self.__dict__.items() if k != "active_dims" or v is not None]
        return self.__class__.__name__ + "(" + ", ".join(args) + ")"

    # -- lowering ---------------------------------------------------------------------------------
    def _user_defined(self):
        """A leaf whose arithmetic is the subclass's own Python `k` (no device kind, or `k` overridden)."""
        return self._kind is None or type(self).k is not Covariance.k

    def _emit(self, cols, leaves, toks):
        """Append this node's leaves / tokens; `cols` are the original column indices visible here.
        Raises NotImplementedError for user-defined leaves (the caller falls back to block evaluation)."""
        if self._user_defined():
            raise NotImplementedError(
                f"Covariance {self.__class__.__name__} defines a Python-level k(): it is evaluated by the binding in "
                "row blocks (BlockCov), not by the device kernel program.")
        dims = compose_active_dims(cols, self.active_dims)
        leaves.append((self._kind, getattr(self, "ls", 1.0), getattr(self, "alpha", 1.0), dims))
        toks.append((_lib.OP_LEAF, len(leaves) - 1, 0.0))

    def _try_program(self, cols):
        """(leaves, toks) when this sub-tree is one device program over the columns `cols`, else None."""
        leaves, toks = [], []
        try:
            self._emit(cols, leaves, toks)
        except NotImplementedError:
            return None
        return (leaves, toks) if _program_fits(leaves, toks) else None

    def lower(self, d):
        """What the C ABI takes for inputs with d feature columns: an mln_kernel_desc (LoweredCov) when the tree is one
        device program, else a BlockCov (values computed block-wise by the binding)."""
        prog = self._try_program(np.arange(d))
        if prog is not None:
            return LoweredCov(*prog)
        return BlockCov(self, d)

    def _block(self, cols, x, y):
        """Kernel values of one block; x, y carry ALL original columns, `cols` the ones visible at this node."""
        prog = self._try_program(cols)
        if prog is not None:
            ctx = _lib.default_context()
            out = ctx.empty((x.shape[0], y.shape[0]))
            ctx._check(ctx.lib.mln_kernel_matrix(ctx.handle, LoweredCov(*prog).ref, x.ctypes.data, x.shape[0],
                                                 y.ctypes.data, y.shape[0], x.shape[1], out.ptr))
            return out
        # user-defined leaf: the subclass's k sees the columns visible here (enclosing active_dims applied, its own
        # selection is its own business, as in the reference where k() calls select_active_dims itself)
        full = len(cols) == x.shape[1] and np.array_equal(cols, np.arange(x.shape[1]))
        xs, ys = (x, y) if full else (np.ascontiguousarray(x[:, cols]), np.ascontiguousarray(y[:, cols]))
        vals = np.asarray(type(self).k(self, xs, ys), dtype=np.float64)
        if vals.shape != (x.shape[0], y.shape[0]):
            raise ValueError(f"{self.__class__.__name__}.k returned shape {vals.shape}, expected {(x.shape[0], y.shape[0])}")
        return np.ascontiguousarray(vals)

    # -- evaluation ---------------------------------------------------------------------------------
    def k(self, x, y):
        x = np.ascontiguousarray(ensure_2d(x), dtype=np.float64)
        y = np.ascontiguousarray(ensure_2d(y), dtype=np.float64)
        return _lib.default_context().kernel_matrix(self.lower(x.shape[1]), x, y)

    def __call__(self, x, y):
        return self.k(x, y)

    def k_grad(self, x):
        """Callable y -> d k(x, y) / d y of shape (n, m, d), zeros on inactive dims (reference
        cov.py k_grad of every kernel, base_cov.py:317-497 for Add/Mul/Pow): the analytic rules are
        evaluated by one HIP kernel over the lowered program instead of nested Python closures."""
        x = np.ascontiguousarray(ensure_2d(x), dtype=np.float64)
        lowered = self.lower(x.shape[1])

        def k_grad(y):
            y = np.ascontiguousarray(ensure_2d(y), dtype=np.float64)
            return _lib.default_context().kernel_grad(lowered, x, y)

        return k_grad

    def diag(self, x):
        """reference base_cov.py:71-93: k(x_i, x_i) for every sample."""
        x = np.ascontiguousarray(ensure_2d(x), dtype=np.float64)
        out = np.empty(x.shape[0])
        step = 4096
        for i0 in range(0, x.shape[0], step):
            blk = x[i0:i0 + step]
            out[i0:i0 + step] = np.diagonal(self.k(blk, blk))
        return out

    def __add__(self, other):
        return Add(self, other)

    def __radd__(self, other):
        return Add(self, other)

    def __mul__(self, other):
        return Mul(self, other)

    def __rmul__(self, other):
        return Mul(self, other)

    def __pow__(self, other):
        return Pow(self, other)

    # -- serialization (reference base_cov.py:103-224) ------------------------------------------------
    def _data_dict(self):
        return {k: make_serializable(v) for k, v in self.__dict__.items()}

    def _metadata(self, module_name):
        from . import __version__
        return {
            "classname": self.__class__.__name__,
            "module_name": module_name,
            "module_version": __version__,
            "serialization_date": datetime.now().isoformat(),
            "python_version": sys.version,
        }

    def _ref_module(self):
        mod = self.__class__.__module__
        if mod.split(".")[0] == __name__.split(".")[0]:
            # serialize under the reference's package so upstream Mellon can load it
            return MELLON_NAME + "." + mod.split(".", 1)[1] if "." in mod else MELLON_NAME
        return mod

    def __getstate__(self):
        return {"type": "mellon.Covariance", "data": self._data_dict(),
                "metadata": self._metadata(self._ref_module())}

    def __setstate__(self, state):
        for name, value in state["data"].items():
            setattr(self, name, deserialize(value))

    def to_json(self):
        return json.dumps(self.__getstate__())

    def to_dict(self):
        return self.__getstate__()

    @classmethod
    def from_json(cls, json_str):
        return cls.from_dict(json.loads(json_str))

    @classmethod
    def from_dict(cls, state):
        if not isinstance(state, dict) or state.get("type") != "mellon.Covariance":
            raise ValueError("The passed dict does not seem to define a covariance kernel.")
        clsname = state["metadata"]["classname"]
        module_name = state["metadata"]["module_name"]
        Sub = _resolve_class(clsname, module_name)
        inst = Sub.__new__(Sub)
        inst.__setstate__(state)
        return inst


def _resolve_class(clsname, module_name):
    if clsname in globals():
        return globals()[clsname]
    from . import cov as _cov
    if hasattr(_cov, clsname) and module_name.split(".")[0] in (MELLON_NAME, __name__.split(".")[0]):
        return getattr(_cov, clsname)
    return getattr(import_module(module_name), clsname)


class CovariancePair(Covariance):
    """reference base_cov.py:227-298."""

    _op = None

    def __init__(self, left, right, active_dims=None):
        super().__init__()
        self.left = left
        self.right = right
        self.active_dims = active_dims

    def _emit(self, cols, leaves, toks):
        cols = compose_active_dims(cols, self.active_dims)
        self.left._emit(cols, leaves, toks)
        if callable(self.right):
            if self._op == _lib.OP_POW:
                raise NotImplementedError("covariance ** covariance is not defined")
            self.right._emit(cols, leaves, toks)
        else:
            toks.append((_lib.OP_CONST, 0, float(self.right)))
        toks.append((self._op, 0, 0.0))

    def _user_defined(self):
        return False

    def _block(self, cols, x, y):
        prog = self._try_program(cols)
        if prog is not None:
            return Covariance._block(self, cols, x, y)
        ctx = _lib.default_context()
        cols = compose_active_dims(cols, self.active_dims)
        left = self.left._block(cols, x, y)
        left = left if isinstance(left, _lib.DeviceArray) else ctx.to_device(left)
        if callable(self.right):
            if self._op == _lib.OP_POW:
                raise NotImplementedError("covariance ** covariance is not defined")
            right = self.right._block(cols, x, y)
            right = right if isinstance(right, _lib.DeviceArray) else ctx.to_device(right)
        else:
            right = float(self.right)
        return ctx.ewise(self._op, left, right)          # in place on the left block

    def __getstate__(self):
        right = self.right.__getstate__() if callable(self.right) else make_serializable(self.right)
        return {"type": "mellon.Covariance", "left_data": self.left.__getstate__(), "right_data": right,
                "active_dims": make_serializable(self.active_dims), "metadata": self._metadata(MELLON_NAME)}

    def __setstate__(self, state):
        if not isinstance(state, dict) or state.get("type") != "mellon.Covariance":
            raise ValueError("The passed dict does not seem to define a covariance kernel.")
        self.left = Covariance.from_dict(state["left_data"])
        rd = state["right_data"]
        if isinstance(rd, dict) and rd.get("type") == "mellon.Covariance":
            self.right = Covariance.from_dict(rd)
        else:
            self.right = deserialize(rd)
        self.active_dims = deserialize(state.get("active_dims", None))


class Add(CovariancePair):
    """reference base_cov.py:301-315."""
    _op = _lib.OP_ADD

    def __repr__(self):
        return "(" + repr(self.left) + " + " + repr(self.right) + ")"


class Mul(CovariancePair):
    """reference base_cov.py:367-381."""
    _op = _lib.OP_MUL

    def __repr__(self):
        return "(" + repr(self.left) + " * " + repr(self.right) + ")"


class Pow(CovariancePair):
    """reference base_cov.py:441-453."""
    _op = _lib.OP_POW

    def __repr__(self):
        return "(" + repr(self.left) + " ** " + repr(self.right) + ")"
