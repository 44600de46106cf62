"""Device eigensolver check: residual, orthogonality, eigenvalues vs LAPACK, time."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mellon_amd import _lib

ctx = _lib.default_context()
rng = np.random.default_rng(0)
for m in [int(a) for a in sys.argv[1:]] or [5, 33, 100, 1000]:
    x = rng.normal(size=(m, 5))
    d2 = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1) if m <= 2000 else None
    if d2 is None:
        xx = (x * x).sum(1)
        d2 = np.maximum(xx[:, None] - 2 * x @ x.T + xx[None, :], 0)
    A = np.exp(-0.5 * d2 / 4.0) + 1e-6 * np.eye(m)     # ExpQuad kernel matrix: huge dynamic range
    t0 = time.perf_counter(); w, V = ctx.eigh(A); t1 = time.perf_counter()
    t2 = time.perf_counter(); w_ref = np.linalg.eigvalsh(A); t3 = time.perf_counter()
    res = np.abs(A @ V - V * w).max() / np.abs(w).max()
    orth = np.abs(V.T @ V - np.eye(m)).max()
    dw = np.abs(w - w_ref).max() / np.abs(w_ref).max()
    print(f"m={m}: sweeps={ctx.last_eigh_sweeps} time={t1 - t0:.3f}s (lapack eigvalsh {t3 - t2:.3f}s) "
          f"resid={res:.2e} orth={orth:.2e} dw={dw:.2e} wmin={w[0]:.3e} wmax={w[-1]:.3e}", flush=True)
