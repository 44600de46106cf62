"""BASELINE.json configs 2, 4, 5 at FULL size on one MI355X: wall-clock + size-independent parity
properties (predict(X) == fit_predict(X); column-wise consistency for the FunctionEstimator)."""
import gc, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, mellon_amd
from mellon_amd import _lib
ctx = _lib.default_context()
out = {}

def relmax(a, b): return float(np.abs(a - b).max() / np.abs(b).max())

which = sys.argv[1:] or ["c2", "c4", "c5"]
if "c2" in which:
    n, d, m = 100_000, 20, 1000
    x = bench.gaussian_mixture(n, d, 2); lm = bench.make_landmarks(x, m, "device", ctx)[0]; xd = ctx.to_device(x); nn = ctx.nn_distances(xd)
    for rep in range(2):
        t0 = time.perf_counter()
        est = mellon_amd.DensityEstimator(cov_func_curry=mellon_amd.cov.ExpQuad, landmarks=lm, nn_distances=nn, check_rank=False)
        dens = est.fit_predict(xd); t1 = time.perf_counter()
    t2 = time.perf_counter(); pred = est.predict(x); t3 = time.perf_counter()
    out["c2"] = dict(n=n, d=d, m=m, kernel="ExpQuad", fit_predict_s=t1 - t0, cells_per_s=n / (t1 - t0),
                     evals=est.loss_func.n_eval, predict_s=t3 - t2, predict_cells_per_s=n / (t3 - t2),
                     predict_eq_fit_predict=relmax(pred, dens))
    del est; gc.collect()
if "c4" in which:
    n, d, m, T = 500_000, 30, 2000, 8
    rng = np.random.default_rng(4)
    xs = bench.gaussian_mixture(n, d, 4); times = np.repeat(np.arange(float(T)), n // T)
    xs = xs + 0.2 * times[:, None]
    xt = np.ascontiguousarray(np.concatenate([xs, times[:, None]], axis=1))
    t0 = time.perf_counter()
    nn = np.empty(n)
    for t in range(T):
        idx = np.flatnonzero(times == t); nn[idx] = ctx.nn_distances(np.ascontiguousarray(xs[idx]))
    t_nn = time.perf_counter() - t0
    ls = float(np.exp(np.log(nn).mean() + 3.0))
    km = xt[rng.choice(n, 20000, replace=False)].copy(); km[:, -1] *= ls / 1.5
    from sklearn.cluster import k_means
    lm = k_means(km, m, n_init=1, random_state=42, max_iter=10, init="random")[0]; lm[:, -1] /= ls / 1.5
    for rep in range(2):
        t0 = time.perf_counter()
        est = mellon_amd.TimeSensitiveDensityEstimator(landmarks=lm, nn_distances=nn, ls_time=1.5, d=d, check_rank=False)
        dens = est.fit_predict(xt); t1 = time.perf_counter()
    k = n
    est.predict(xt[:1000])
    t2 = time.perf_counter(); pred = est.predict(xt[:k]); t3 = time.perf_counter()
    xq = ctx.to_device(xt)
    t4 = time.perf_counter(); pred_dev = est.predict(xq); t5 = time.perf_counter()
    out["c4"] = dict(n=n, d=d, m=m, timepoints=T, nn_within_time_s=t_nn, fit_predict_s=t1 - t0,
                     cells_per_s=n / (t1 - t0), evals=est.loss_func.n_eval, predict_cells_per_s=k / (t3 - t2),
                     predict_cells_per_s_resident_queries=k / (t5 - t4),
                     predict_eq_fit_predict=relmax(pred, dens[:k]), cov=repr(est.cov_func))
    del est; gc.collect()
if "c5" in which:
    n, d, m, p = 200_000, 50, 2000, 2000
    rng = np.random.default_rng(5)
    x = bench.gaussian_mixture(n, d, 5); W = rng.normal(size=(d, p)) / np.sqrt(d)
    y = np.sin(x @ W) + 0.1 * rng.normal(size=(n, p))
    lm = bench.make_landmarks(x, m, "device", ctx)[0]; nn = ctx.nn_distances(x)
    t0 = time.perf_counter()
    est = mellon_amd.FunctionEstimator(sigma=0.1, landmarks=lm, nn_distances=nn)
    est.fit(x, y); t1 = time.perf_counter()
    pred = est.predict(x); t2 = time.perf_counter()
    one = mellon_amd.FunctionEstimator(sigma=0.1, landmarks=lm, nn_distances=nn).fit(x, y[:, 7]).predict(x[:20000])
    out["c5"] = dict(n=n, d=d, m=m, p=p, fit_s=t1 - t0, predict_s=t2 - t1, cells_per_s_fit_predict=n / (t2 - t0),
                     column_consistency=relmax(pred[:20000, 7], one), resid_std=float(np.std(pred - y)))
print(json.dumps(out, indent=1))
