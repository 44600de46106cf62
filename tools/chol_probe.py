"""chol(m) through the C-ABI (mln_chol_lower): backward error against the input and, under `rocprofv3 --kernel-trace --stats`,
the per-launch time of k_potrf128 and the GEMMs of the chain.   python tools/chol_probe.py [m]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from mellon_amd import _lib
ctx = _lib.default_context()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
rng = np.random.default_rng(0)
B = rng.normal(size=(m, m // 2))
A = B @ B.T + 1e-3 * m * np.eye(m)           # condition ~1e3-1e4
for rep in range(3):
    t0 = time.perf_counter()
    L = ctx.chol_lower(A)
    dt = time.perf_counter() - t0
R = L @ L.T - A
print(f"chol({m}): {dt * 1e3:.1f} ms incl. copies; backward error {np.abs(R).max() / np.abs(A).max():.2e}; "
      f"vs numpy {np.abs(L - np.linalg.cholesky(A)).max() / np.abs(L).max():.2e}")
