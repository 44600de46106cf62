"""Golden fixture for the C3-shaped parity gate ("C3 subsample, n = 1e5"):
1e5 cells x 50 dims (gaussian_mixture seed 3), 5000 landmarks (a seeded random subset of the cells, so
that nothing but seeds has to be stored), Matern52, exact 1-NN distances, all other heuristics default.
Generated HERE by the oracle (the reference cannot be imported: SURVEY.md S8c).  The strictly convex MAP
problem is solved to gtol 1e-9 on the Ridge-preconditioned variable (same optimum as the plain variable,
~40 instead of ~1300 passes over the 4 GB factor).  Stored: every 50th log-density, z*, mu, ls, loss.

    python tests/golden/make_c3_subsample.py        # ~10-15 min, ~20 GB RAM
"""
import os
import sys
import time

import numpy as np
import scipy.linalg as sla
import scipy.optimize as so

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mellon_oracle as mo  # noqa: E402

N, D, M, SEED, LM_SEED, KEEP = 100_000, 50, 5000, 3, 33, 50


def inputs():
    x = mo.gaussian_mixture(N, D, seed=SEED)
    idx = np.sort(np.random.default_rng(LM_SEED).choice(N, M, replace=False))
    return x, idx


def main():
    t0 = time.time()
    x, idx = inputs()
    lm = x[idx]
    nn = mo.exact_nn_distances(x)
    print("nn done", time.time() - t0, flush=True)
    mu, ls = mo.compute_mu(nn, D), mo.compute_ls(nn)
    cov = mo.Matern52(ls)
    Lp = mo.full_rank(lm, cov)
    L = mo.standard_low_rank(x, cov, lm, Lp=Lp)
    print("L done", time.time() - t0, flush=True)
    V, Vdr = mo.nn_likelihood_constants(nn, D)
    G = L.T @ L
    G[np.diag_indices_from(G)] += 1.0
    C = np.linalg.cholesky(G)
    z0 = sla.cho_solve((C, True), L.T @ (mo.mle(nn, D) - mu))
    print("ridge done", time.time() - t0, flush=True)
    n_eval = [0]

    def fun(u):
        n_eval[0] += 1
        z = sla.solve_triangular(C, u, lower=True, trans="T")
        loss, g = mo.loss_and_grad(z, L, mu, V, Vdr)
        return loss, sla.solve_triangular(C, g, lower=True)

    res = so.minimize(fun, C.T @ z0, jac=True, method="L-BFGS-B",
                      options=dict(maxcor=30, ftol=0.0, gtol=1e-9, maxiter=2000, maxfun=4000))
    z = sla.solve_triangular(C, res.x, lower=True, trans="T")
    loss, g = mo.loss_and_grad(z, L, mu, V, Vdr)
    f = L @ z + mu
    print("solve done", time.time() - t0, "evals", n_eval[0], "max|grad_z|", np.abs(g).max(), "loss", loss, flush=True)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c3_sub_density.npz"),
                        n=N, dims=D, m=M, seed=SEED, landmark_seed=LM_SEED, keep_every=KEEP,
                        log_density_sub=f[::KEEP], pre_transformation=z, mu=mu, ls=ls, loss=loss,
                        nn_sub=nn[::KEEP], grad_max=np.abs(g).max())


if __name__ == "__main__":
    main()
