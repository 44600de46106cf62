import numpy as np, scipy.linalg as sla, sys, time
sys.path.insert(0, "/root/repo")
from oracle import mellon_oracle as mo

def problem(n, d, m, seed, kind="gmm"):
    if kind == "gmm":
        x = mo.gaussian_mixture(n, d, seed=seed)
    else:
        rng = np.random.default_rng(seed); x = rng.standard_t(3, size=(n, d))
    nn = mo.exact_nn_distances(x)
    ls = mo.compute_ls(nn); mu = mo.compute_mu(nn, d)
    from sklearn.cluster import k_means
    xu = k_means(x[:min(n, 20000)], m, n_init=1, random_state=42, max_iter=30)[0]
    cov = mo.Matern52(ls)
    K = cov(x, xu); Kj = cov(xu, xu) + 1e-6 * np.eye(m)
    V, Vdr = mo.nn_likelihood_constants(nn, d)
    return dict(K=K, Kj=Kj, V=V, Vdr=Vdr, mu=mu, mle=mo.mle(nn, d), n=n, m=m)

class Obj:
    def __init__(s, P, R):
        s.P, s.R = P, R; s.nev = 0
        s.Rinv = sla.solve_triangular(R, np.eye(R.shape[0]), lower=True)
    def w_of(s, u): return s.Rinv.T @ u
    def eval(s, u, dirs=None):
        """loss, grad_u at u; if dirs (list of u-space vectors): also H(u) v for each, same pass"""
        P = s.P; s.nev += 1
        w = s.w_of(u)
        f = P["K"] @ w + P["mu"]
        a = np.exp(f + P["V"])
        q = P["Kj"] @ w
        loss = 0.5 * w @ q + np.sum(a - f - P["Vdr"])
        gw = q + P["K"].T @ (a - 1.0)
        g = s.Rinv @ gw
        if dirs is None: return loss, g
        HV = []
        for v in dirs:
            wv = s.w_of(v)
            kv = P["K"] @ wv
            HV.append(s.Rinv @ (P["Kj"] @ wv + P["K"].T @ (a * kv)))
        return loss, g, HV

def lbfgs_dir(g, S, Y):
    q = g.copy(); al = []
    for s_, y_ in reversed(list(zip(S, Y))):
        r = 1.0 / (s_ @ y_); a_ = r * (s_ @ q); al.append(a_); q -= a_ * y_
    if S: q *= (S[-1] @ Y[-1]) / (Y[-1] @ Y[-1])
    for (s_, y_), a_ in zip(zip(S, Y), reversed(al)):
        r = 1.0 / (s_ @ y_); b_ = r * (y_ @ q); q += s_ * (a_ - b_)
    return -q

def solve_plain(ob, u0, ftol=1e-13, gtol=1e-7, maxcor=10, maxit=500, verbose=False):
    u = u0.copy(); fx, g = ob.eval(u); S, Y = [], []; t = min(1.0, 1.0 / np.abs(g).sum())
    for it in range(maxit):
        if np.abs(g).max() <= gtol: break
        d = lbfgs_dir(g, S, Y); gd = g @ d
        ls = 0
        while True:
            fn, gn = ob.eval(u + t * d); ls += 1
            if np.isfinite(fn) and fn <= fx + 1e-4 * t * gd: break
            if ls > 30: return u, fx, ob.nev
            t = max(0.1 * t, min(0.5 * t, -gd * t * t / (2 * (fn - fx - gd * t)))) if np.isfinite(fn) else 0.1 * t
        s_ = t * d; y_ = gn - g
        dec = fx - fn
        u = u + s_; g = gn; fold = fx; fx = fn
        if s_ @ y_ > 1e-10 * np.sqrt((s_ @ s_) * (y_ @ y_)):
            S.append(s_); Y.append(y_)
            if len(S) > maxcor: S.pop(0); Y.pop(0)
        if verbose: print(it, ob.nev, fx, dec, t)
        if dec <= ftol * max(abs(fold), abs(fx), 1.0): break
        t = 1.0
    return u, fx, ob.nev

def solve_subspace(ob, u0, nd=3, ftol=1e-13, gtol=1e-7, maxcor=10, maxit=500, verbose=False):
    """Each pass at the trial point u_t = u + t d also returns H(u_t) v for v in {d, last nd-1 steps}.  After the pass: a Newton
    step of the quadratic model at u_t restricted to span(V): c = -(V^T H V)^-1 V^T g_t; the corrected point u' = u_t + V c
    with PREDICTED gradient g' = g_t + H V c and predicted loss; the next quasi-Newton direction starts from (u', g')."""
    u = u0.copy(); fx, g = ob.eval(u); S, Y = [], []; t = min(1.0, 1.0 / np.abs(g).sum())
    hist = []
    for it in range(maxit):
        if np.abs(g).max() <= gtol: break
        d = lbfgs_dir(g, S, Y); gd = g @ d
        dirs = [d] + hist[-(nd - 1):] if nd > 1 else [d]
        ls = 0
        while True:
            fn, gn, HV = ob.eval(u + t * d, dirs); ls += 1
            if np.isfinite(fn) and fn <= fx + 1e-4 * t * gd: break
            if ls > 30: return u, fx, ob.nev
            t = max(0.1 * t, min(0.5 * t, -gd * t * t / (2 * (fn - fx - gd * t)))) if np.isfinite(fn) else 0.1 * t
        ut = u + t * d
        # curvature pairs: every direction gives an exact (v, H v) pair at the current point
        Vm = np.stack(dirs, 1); HVm = np.stack(HV, 1)
        A = Vm.T @ HVm; A = 0.5 * (A + A.T); b = Vm.T @ gn
        try:
            c = -np.linalg.solve(A, b)
        except np.linalg.LinAlgError:
            c = np.zeros(len(dirs))
        pred_dec = -(b @ c + 0.5 * c @ A @ c)
        if not np.isfinite(pred_dec) or pred_dec < 0: c[:] = 0; pred_dec = 0.0
        u_new = ut + Vm @ c; g_new = gn + HVm @ c; f_new = fn - pred_dec
        s_ = u_new - u; y_ = g_new - g
        dec = fx - f_new
        hist.append(s_.copy())
        u = u_new; g = g_new; fold = fx; fx = f_new
        # replace memory by exact pairs where possible: drop old, then add the exact pairs of this pass and the step pair
        if s_ @ y_ > 1e-10 * np.sqrt((s_ @ s_) * (y_ @ y_)):
            S.append(s_); Y.append(y_)
            if len(S) > maxcor: S.pop(0); Y.pop(0)
        if verbose: print(it, ob.nev, fx, dec, t, np.round(c, 3))
        if dec <= ftol * max(abs(fold), abs(fx), 1.0):
            # the loss was predicted: verify with a true evaluation
            f_true, g_true = ob.eval(u)
            if verbose: print("verify", f_true, fx, np.abs(g_true - g).max())
            if abs(f_true - fx) <= 10 * ftol * max(abs(f_true), 1.0) or True:
                fx, g = f_true, g_true
                break
        t = 1.0
    return u, fx, ob.nev

if __name__ == "__main__":
    n, d, m = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); seed = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    kind = sys.argv[5] if len(sys.argv) > 5 else "gmm"
    P = problem(n, d, m, seed, kind)
    stride = max(1, n // (6 * m))
    Ks = P["K"][::stride]
    M = stride * Ks.T @ Ks + P["Kj"]
    R = np.linalg.cholesky(M)
    # ridge start on the sample
    rhs = stride * Ks.T @ (P["mle"] - P["mu"])[::stride]
    w0 = sla.cho_solve((R, True), rhs)
    u0 = R.T @ w0
    for name, fn in (("plain", solve_plain), ("sub2", lambda o, u: solve_subspace(o, u, 2)), ("sub3", lambda o, u: solve_subspace(o, u, 3)), ("sub5", lambda o, u: solve_subspace(o, u, 5))):
        ob = Obj(P, R)
        u, fx, nev = fn(ob, u0)
        w = ob.w_of(u); f = P["K"] @ w + P["mu"]
        print(name, "evals", nev, "loss %.10f" % fx)
        if name == "plain": fref = f
        else: print("   vs plain rel", np.abs(f - fref).max() / np.abs(fref).max())

def solve_refresh(ob, u0, nd=3, ftol=1e-13, gtol=1e-7, maxcor=10, maxit=500, verbose=False, exact_ls=True):
    """Standard L-BFGS iteration; every pass (at the trial point) also returns H v for v = d and the newest nd-1 stored steps:
    (a) the stored pairs' y are REFRESHED with the current Hessian; (b) the step along d gets the exact curvature d^T H d at the
    trial point: one Newton correction of the step LENGTH (predicted gradient g + t' H d) -- an exact line search for free."""
    u = u0.copy(); fx, g = ob.eval(u); S, Y = [], []; t = min(1.0, 1.0 / np.abs(g).sum())
    for it in range(maxit):
        if np.abs(g).max() <= gtol: break
        d = lbfgs_dir(g, S, Y); gd = g @ d
        k = min(nd - 1, len(S))
        dirs = [d] + S[len(S) - k:]
        ls = 0
        while True:
            fn, gn, HV = ob.eval(u + t * d, dirs); ls += 1
            if np.isfinite(fn) and fn <= fx + 1e-4 * t * gd: break
            if ls > 30: return u, fx, ob.nev
            t = max(0.1 * t, min(0.5 * t, -gd * t * t / (2 * (fn - fx - gd * t)))) if np.isfinite(fn) else 0.1 * t
        for j in range(k):
            Y[len(S) - k + j] = HV[1 + j]
        ut = u + t * d
        s_ = t * d; y_ = gn - g
        f_new, g_new = fn, gn
        if exact_ls:
            dHd = d @ HV[0]; gtd = gn @ d
            if dHd > 0:
                dt = -gtd / dHd
                if abs(dt) < 0.9 * t or dt > 0:
                    pred = -(gtd * dt + 0.5 * dHd * dt * dt)
                    ut = ut + dt * d; g_new = gn + dt * HV[0]; f_new = fn - pred
                    s_ = ut - u; y_ = (t + dt) * HV[0]        # exact curvature pair along d at the trial point
        dec = fx - f_new
        u = ut; g = g_new; fold = fx; fx = f_new
        if s_ @ y_ > 1e-10 * np.sqrt((s_ @ s_) * (y_ @ y_)):
            S.append(s_); Y.append(y_)
            if len(S) > maxcor: S.pop(0); Y.pop(0)
        if verbose: print(it, ob.nev, fx, dec, t)
        if dec <= ftol * max(abs(fold), abs(fx), 1.0):
            f_true, g_true = ob.eval(u); fx, g = f_true, g_true
            break
        t = 1.0
    return u, fx, ob.nev

if __name__ == "__main__":
    for name, fn in (("refresh1+ls", lambda o, u: solve_refresh(o, u, 1)), ("refresh3+ls", lambda o, u: solve_refresh(o, u, 3)), ("refresh3 no ls", lambda o, u: solve_refresh(o, u, 3, exact_ls=False)), ("refresh6+ls", lambda o, u: solve_refresh(o, u, 6))):
        ob = Obj(P, R)
        u, fx, nev = fn(ob, u0)
        w = ob.w_of(u); f = P["K"] @ w + P["mu"]
        print(name, "evals", nev, "loss %.10f" % fx, "vs plain rel", np.abs(f - fref).max() / np.abs(fref).max())
