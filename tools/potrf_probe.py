"""Phase timing inside k_potrf128 (the library must be built with -DMLN_POTRF_TIMING in _build.FLAGS): global load,
the eight micro-steps, store, inverse; shader clock.  Round 3: 66 -> 50 us per 128-block (batched loads, LDS broadcasts
instead of ~820 v_readlane with spilled scalar registers, software-pipelined pivots)."""
import sys, ctypes as C, numpy as np
sys.path.insert(0, ".")
from mellon_amd import _lib
ctx = _lib.default_context()
rng = np.random.default_rng(0)
m = 2000
B = rng.normal(size=(m, m)); A = B @ B.T + m * np.eye(m)
for _ in range(3): L = ctx.chol_lower(A)
ts = (C.c_longlong * 64)()
print("rc", ctx.lib.mln_diag_potrf_times(ts))
t = np.array(ts[:23], dtype=np.int64)
cc = np.array(ts[32:64], dtype=np.int64)
print('p=3 wave 0: trailing tile + loads', cc[24]-cc[23], 'pivot loop', cc[25]-cc[24], 'final writes', cc[26]-cc[25], 'cycles; step', cc[4+2*3]-cc[3+2*3])
us = (t - t[0]) / 100.0          # wall_clock64: 100 MHz
print("load done", us[1], "diag0", us[2])
for p in range(8):
    print("p", p, "panel done", us[3 + 2 * p], "trailing+diag done", us[4 + 2 * p] if p < 7 else "-")
print("loop end", us[20], "store done", us[21], "inverse done", us[22])
c = np.array(ts[32:55], dtype=np.int64)
print("shader clock over the kernel: %.0f MHz" % ((c[22] - c[0]) / ((t[22] - t[0]) / 100.0)))
print("cycles per diag_tile+trailing (p=3):", c[4 + 2 * 3] - c[3 + 2 * 3])
