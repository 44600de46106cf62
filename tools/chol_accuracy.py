import sys, numpy as np, scipy.linalg as sla
sys.path.insert(0, ".")
from mellon_amd import _lib
ctx = _lib.default_context()
rng = np.random.default_rng(1)
for m, cond in ((500, 1e2), (2000, 1e6), (3000, 1e10)):
    Q, _ = np.linalg.qr(rng.normal(size=(m, m)))
    A = (Q * np.logspace(0, -np.log10(cond), m)) @ Q.T; A = 0.5 * (A + A.T)
    L = ctx.chol_lower(A); Lr = sla.cholesky(A, lower=True)
    print(m, cond, "rel diff vs LAPACK %.2e" % (np.abs(L - Lr).max() / np.abs(Lr).max()), "backward %.2e" % (np.abs(L @ L.T - A).max() / np.abs(A).max()), "LAPACK backward %.2e" % (np.abs(Lr @ Lr.T - A).max() / np.abs(A).max()))
