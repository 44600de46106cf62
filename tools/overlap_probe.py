"""Do two streams (two contexts, two host threads) overlap a VALU-bound kernel-matrix pass with a
latency-bound Cholesky / an MFMA-bound GEMM chain on one MI355X?"""
import sys, time, threading
sys.path.insert(0, ".")
import numpy as np
from mellon_amd import _lib, cov
import bench

c1 = _lib.Context(0); c2 = _lib.Context(0)
n, d, m = 1_000_000, 50, 5000
x = bench.gaussian_mixture(n, d, 3); lm = x[:m].copy()
xd = c1.to_device(x)
k = cov.Matern52(20.0)
desc = k.lower(d)
A = c2.kernel_matrix(desc, lm, lm) + 1e-6 * np.eye(m)

def job_k():
    f = c1.fit_prepare(desc, xd, lm, 1e-6, implicit=True); f.close()
def job_chol():
    c2.chol_lower(A)
for name, jobs in (("K-fit (incl. its own chol)", [job_k]), ("chol", [job_chol]), ("both", [job_k, job_chol])):
    for rep in range(3):
        t0 = time.perf_counter()
        th = [threading.Thread(target=j) for j in jobs]
        [t.start() for t in th]; [t.join() for t in th]
        dt = time.perf_counter() - t0
    print(f"{name}: {dt*1e3:.1f} ms", flush=True)
