"""CPU prototype of the MAP solve's iteration path (NumPy; not part of the product, not used by any test).

Questions it answers before anything is written in HIP: how many passes over the n x m buffer does the preconditioned
L-BFGS of csrc/solver.hip need when the preconditioner is rebuilt ONCE from the a-weighted sampled Gram
(a = exp(f + V) at an intermediate point: the MAP Hessian there instead of the Ridge matrix), and when does the
rebuild pay.  Usage:  python tools/precond_experiment.py [n] [m] [d] [seed]
"""
import sys
import time

import numpy as np
import scipy.linalg as sl

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import mellon_oracle as mo  # noqa: E402


def setup(n, m, d, seed, kern="Matern52"):
    x = mo.gaussian_mixture(n, d, seed)
    from sklearn.neighbors import NearestNeighbors
    nn = NearestNeighbors(n_neighbors=2, algorithm="brute").fit(x).kneighbors(x)[0][:, 1]
    nn = mo.validate_nn_distances(nn)
    mu, ls = mo.compute_mu(nn, d), mo.compute_ls(nn)
    from sklearn.cluster import k_means
    xu = k_means(x[: min(n, 20 * m)], m, n_init=1, random_state=42)[0]
    cov = getattr(mo, kern)(ls)
    Lp = mo.full_rank(xu, cov)
    L = mo.standard_low_rank(x, cov, xu, Lp=Lp)
    V, Vdr = mo.nn_likelihood_constants(nn, d)
    t = mo.mle(nn, d) - mu
    return dict(L=L, V=V, Vdr=Vdr, mu=mu, t=t, n=n, m=m)


class Problem:
    def __init__(self, P):
        self.__dict__.update(P)
        self.passes = 0.0

    def eval_z(self, z, rows=None, scale=1.0):
        """loss, grad_z, f; rows: subsample (cost counted proportionally)."""
        L = self.L if rows is None else self.L[rows]
        V = self.V if rows is None else self.V[rows]
        Vdr = self.Vdr if rows is None else self.Vdr[rows]
        self.passes += L.shape[0] / self.n
        f = L @ z + self.mu
        a = np.exp(f + V)
        loss = 0.5 * z @ z + 0.5 * self.m * np.log(2 * np.pi) - scale * np.sum(f + Vdr - a)
        g = z + scale * (L.T @ (a - 1.0))
        return loss, g, f, a


def chol_precond(G):
    C = sl.cholesky(G, lower=True)
    return C


def lbfgs(prob, z0, C, ftol=1e-13, gtol=1e-7, maxcor=10, maxiter=500, rows=None, scale=1.0, boost=0.15, stop_rel=None,
          verbose=False, pairs=None):
    """Armijo L-BFGS on u (z = C^-T u), as csrc/solver.hip.  Returns z, info."""
    to_z = lambda u: sl.solve_triangular(C, u, lower=True, trans="T")
    to_gu = lambda gz: sl.solve_triangular(C, gz, lower=True)
    u = C.T @ z0
    fx, gz, f, a = prob.eval_z(to_z(u), rows, scale)
    g = to_gu(gz)
    S, Y = ([], []) if pairs is None else pairs
    it = n_eval = 1
    t0 = 1.0
    hist = [fx]
    while it < maxiter:
        if np.abs(g).max() <= gtol:
            break
        q = g.copy()
        al = []
        for s, y in zip(reversed(S), reversed(Y)):
            r = 1.0 / (s @ y)
            a_ = r * (s @ q)
            al.append(a_)
            q -= a_ * y
        if S:
            q *= (S[-1] @ Y[-1]) / (Y[-1] @ Y[-1])
        for (s, y), a_ in zip(zip(S, Y), reversed(al)):
            r = 1.0 / (s @ y)
            b = r * (y @ q)
            q += s * (a_ - b)
        dvec = -q
        gd = g @ dvec
        t = t0 if S else min(1.0, 1.0 / np.abs(g).sum())
        ls = 0
        while True:
            un = u + t * dvec
            fn, gzn, fnew, anew = prob.eval_z(to_z(un), rows, scale)
            n_eval += 1
            ls += 1
            if np.isfinite(fn) and fn <= fx + 1e-4 * t * gd:
                break
            if np.isfinite(fn) and abs(fn - fx) <= ftol * max(abs(fx), abs(fn), 1.0):
                return to_z(u), dict(n_eval=n_eval, it=it, hist=hist, a=a, f=f, pairs=(S, Y), fx=fx)
            if ls >= 30:
                return to_z(u), dict(n_eval=n_eval, it=it, hist=hist, a=a, f=f, pairs=(S, Y), fx=fx, fail=True)
            if np.isfinite(fn):
                tq = -gd * t * t / (2.0 * (fn - fx - gd * t))
                t = min(max(tq, 0.1 * t), 0.5 * t)
            else:
                t *= 0.1
        gn = to_gu(gzn)
        s, y = un - u, gn - g
        sl_ = gn @ dvec
        t0 = min(2 * t, 16.0) if (boost > 0 and ls == 1 and t >= 1.0 and sl_ / gd > boost and (fx - fn) > 0.15 * abs(fx)) else 1.0
        f_old = fx
        u, g, fx, f, a = un, gn, fn, fnew, anew
        if s @ y > 1e-10 * np.sqrt((s @ s) * (y @ y)):
            S.append(s); Y.append(y)
            if len(S) > maxcor:
                S.pop(0); Y.pop(0)
        it += 1
        hist.append(fx)
        if verbose:
            print(f"   it {it} evals {n_eval} loss {fx:.10g} rel dec {(f_old - fx) / max(abs(fx), 1):.2e}")
        if (f_old - fx) <= ftol * max(abs(f_old), abs(fx), 1.0):
            break
        if stop_rel is not None and it >= 4 and (f_old - fx) <= stop_rel * max(abs(f_old), abs(fx), 1.0):
            break
    return to_z(u), dict(n_eval=n_eval, it=it, hist=hist, a=a, f=f, pairs=(S, Y), fx=fx)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    d = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    t0 = time.time()
    kern = sys.argv[5] if len(sys.argv) > 5 else "Matern52"
    P = setup(n, m, d, seed, kern)
    print(f"setup n={n} m={m} d={d}: {time.time() - t0:.1f} s", flush=True)
    L, t = P["L"], P["t"]
    I = np.eye(m)

    def sample(k, phase=0):
        s = max(1, n // (k * m))
        return np.arange(phase % s, n, s), s

    def ridge(rows, s, w=None):
        Ls = L[rows]
        G = s * (Ls.T @ (Ls if w is None else Ls * w[:, None])) + I
        return chol_precond(G)

    results = {}
    # reference optimum
    prob = Problem(P)
    rows12, s12 = sample(12)
    C12 = ridge(rows12, s12)
    rhs = L.T @ t
    z0 = sl.cho_solve((C12, True), rhs)
    z_ref, info = lbfgs(prob, z0, C12, ftol=1e-15, gtol=1e-10, maxiter=2000)
    f_ref = L @ z_ref + P["mu"]
    print(f"reference optimum: {info['n_eval']} evals, loss {info['fx']:.12g}", flush=True)

    def report(name, prob, z, extra_cost=0.0, evals=None):
        err = np.abs(L @ z + P["mu"] - f_ref).max() / np.abs(f_ref).max()
        print(f"{name:58s} passes {prob.passes:6.2f} (+{extra_cost:.1f} rebuild-equiv) evals {evals} err {err:.2e}", flush=True)

    for k in (12, 32):
        prob = Problem(P)
        rows, s = sample(k)
        C = ridge(rows, s)
        z0 = sl.cho_solve((C, True), rhs)
        z, info = lbfgs(prob, z0, C)
        report(f"baseline Ridge precond {k}m rows", prob, z, evals=info["n_eval"])
    prob = Problem(P)
    Call = ridge(np.arange(n), 1)
    z0 = sl.cho_solve((Call, True), rhs)
    z, info = lbfgs(prob, z0, Call)
    report("baseline Ridge precond ALL rows", prob, z, evals=info["n_eval"])

    rng = np.random.default_rng(0)

    def importance(a, k, power=1.0):
        """rows drawn with p_i = min(1, c a_i^power), expected count k m; weights a_i / p_i"""
        q = a ** power
        c = k * m / q.sum()
        for _ in range(20):                      # fix c so that sum min(1, c q) = k m
            p = np.minimum(1.0, c * q)
            c *= k * m / p.sum()
        p = np.minimum(1.0, c * q)
        pick = rng.random(a.shape[0]) < p
        return np.nonzero(pick)[0], a[pick] / p[pick]

    def weighted(rows, w):
        Ls = L[rows]
        return chol_precond(Ls.T @ (Ls * w[:, None]) + I)

    rows, s = sample(12)
    z0 = sl.cho_solve((C12, True), rhs)

    def stats(a):
        srt = np.sort(a)[::-1]
        cs = np.cumsum(srt) / srt.sum()
        return "a: mean %.3g median %.3g max %.3g top1%%/5%%/20%% share %.2f %.2f %.2f" % (a.mean(), np.median(a), a.max(), cs[n // 100], cs[n // 20], cs[n // 5])

    for sub_tol in (None, 1e-3):
        for thr in (0.01, 0.003):
            prob = Problem(P)
            if sub_tol is None:
                z1 = z0; sub = 0.0
            else:
                z1, i1 = lbfgs(prob, z0, C12, rows=rows, scale=float(s), ftol=sub_tol)
                sub = prob.passes
            zf, i_f = lbfgs(prob, z1, C12, stop_rel=thr)
            a1 = i_f["a"]
            r2, w2 = importance(a1, 12)
            C2 = weighted(r2, w2)
            z2, i2 = lbfgs(prob, zf, C2)
            report(f"sub tol {sub_tol} ({sub:.2f}) + full rel<{thr} ({i_f['n_eval']}) + importance + tail {i2['n_eval']}", prob, z2, evals=i_f["n_eval"] + i2["n_eval"])
            if thr == 0.01:
                print("    ", stats(a1), flush=True)


if __name__ == "__main__":
    main()
