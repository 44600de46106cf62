"""Host-to-device copy rate of the C3 cells (0.4 GB) from pageable and from page-locked (mln_host_register) memory, whole and in
16 chunks: the ceiling of what the fit's chunked upload can hide under its kernel-matrix pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mellon_amd import _lib
ctx = _lib.default_context()
x = np.random.default_rng(0).normal(size=(1_000_000, 50))
d = ctx.empty(x.shape)
def rate(label):
    for rep in range(3):
        t0 = time.perf_counter()
        ctx._check(ctx.lib.mln_memcpy(ctx.handle, d.ptr, x.ctypes.data, x.nbytes))
        dt = time.perf_counter() - t0
    print(f"{label:28s} {1e3 * dt:7.2f} ms  {x.nbytes / dt / 1e9:6.1f} GB/s", flush=True)
    rows = x.shape[0] // 16
    for rep in range(2):
        t0 = time.perf_counter()
        for c in range(16):
            ctx._check(ctx.lib.mln_memcpy(ctx.handle, d.ptr + c * rows * 400, x.ctypes.data + c * rows * 400, rows * 400))
        dt = time.perf_counter() - t0
    print(f"{label + ', 16 chunks':28s} {1e3 * dt:7.2f} ms  {x.nbytes / dt / 1e9:6.1f} GB/s", flush=True)
rate("pageable")
with ctx.pinned(x):
    rate("page-locked (registered)")
out = np.empty(1_000_000)
dv = ctx.empty((1_000_000,))
for rep in range(3):
    t0 = time.perf_counter(); ctx._check(ctx.lib.mln_memcpy(ctx.handle, out.ctypes.data, dv.ptr, out.nbytes)); dt = time.perf_counter() - t0
print(f"D2H 8 MB pageable            {1e3 * dt:7.2f} ms  {out.nbytes / dt / 1e9:6.1f} GB/s")
