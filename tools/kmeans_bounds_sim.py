"""CPU simulation (NumPy, 1e5 cells x 50 dims -> 500 centres, the cells-per-centre ratio of C3) of how much of Lloyd's tail a
lower bound per GROUP of centres (Yinyang) would leave to search, against Hamerly's single bound (what mln_kmeans does since
round 4b): per sweep >= 25, the fraction of cells touched and the fraction of (cell, group) pairs searched.
   groups = k-means of the centres (16):  cells 0.21, pairs 0.014
   groups = index blocks (16):            cells 0.65, pairs 0.083
   one group (Hamerly):                   cells 0.40, pairs 0.40      (the device run measures 0.39)
"""
import sys
sys.path.insert(0, ".")
import numpy as np, bench
from sklearn.cluster import KMeans
n, m, d = 100_000, 500, 50
x = bench.gaussian_mixture(n, d, 3)
km = KMeans(m, n_init=1, random_state=0, tol=1e-4, max_iter=300).fit(x[::8])
c = km.cluster_centers_.copy()
tol = 1e-4 * x.var(axis=0).mean()
xx = (x * x).sum(1)
for G, how in ((16, "kmeans"), (16, "index"), (1, "hamerly")):
    c = km.cluster_centers_.copy()
    if how == "kmeans":
        grp = KMeans(G, n_init=1, random_state=1).fit(c).labels_
    elif how == "index":
        grp = np.arange(m) * G // m
    else:
        grp = np.zeros(m, int)
    prev_lab = None
    lbg = None; searched_pairs = 0; total_pairs = 0; searched_pts = 0; total_pts = 0
    for it in range(300):
        D = np.sqrt(np.maximum(xx[:, None] - 2 * x @ c.T + (c * c).sum(1)[None, :], 0))
        lab = D.argmin(1)
        ub = D[np.arange(n), lab]
        Dm = D.copy(); Dm[np.arange(n), lab] = np.inf
        true_lbg = np.stack([Dm[:, grp == g].min(1) if np.any(grp == g) else np.full(n, np.inf) for g in range(G)], 1)
        if lbg is None:
            lbg = true_lbg.copy()
        else:
            need = lbg < ub[:, None]              # groups that must be searched for this point
            if it >= 25:
                searched_pairs += need.sum(); total_pairs += need.size
                searched_pts += need.any(1).sum(); total_pts += n
            lbg = np.where(need, true_lbg, lbg)   # refreshed where searched (exact), else keep the decayed bound
            ch = np.nonzero(lab != prev_lab)[0]
            if ch.size:
                g_old = grp[prev_lab[ch]]
                lbg[ch, g_old] = np.minimum(lbg[ch, g_old], D[ch, prev_lab[ch]])
            # sanity: bounds valid
            assert np.all(lbg <= true_lbg + 1e-9)
        prev_lab = lab
        cn = np.zeros_like(c); cnt = np.bincount(lab, minlength=m)
        np.add.at(cn, lab, x); nz = cnt > 0
        cn[nz] /= cnt[nz, None]; cn[~nz] = c[~nz]
        delta = np.sqrt(((cn - c) ** 2).sum(1))
        gmax = np.array([delta[grp == g].max() if np.any(grp == g) else 0.0 for g in range(G)])
        lbg = lbg - gmax[None, :]
        shift = (delta ** 2).sum(); c = cn
        if shift <= tol: break
    print(how, G, "sweeps", it + 1, "tail (it>=25): points touched %.3f, (point,group) pairs searched %.3f" % (searched_pts / max(total_pts, 1), searched_pairs / max(total_pairs, 1)))
