"""Stage-level timing probe of the hot path at bench sizes (synthetic nn distances)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mellon_amd import _lib, cov

def main(n, d, m, kern="Matern52", evals=20):
    rng = np.random.default_rng(0)
    ctx = _lib.default_context()
    print(ctx.device_info())
    x = rng.normal(size=(n, d)) * 2.0
    xu = x[rng.choice(n, m, replace=False)] + 0.05 * rng.normal(size=(m, d))
    nn = rng.uniform(0.5, 1.5, size=n) * np.sqrt(d) * 0.3
    ls = float(np.exp(np.log(nn).mean() + 3.0))
    c = getattr(cov, kern)(ls)
    xd = ctx.to_device(x)
    t0 = time.time()
    fit = ctx.fit_prepare(c.lower(d), xd, xu, 1e-6)
    t1 = time.time()
    print(f"fit_prepare {t1-t0:.3f}s", {k: round(v, 4) for k, v in fit.stage_times().items()})
    from scipy.special import gammaln
    const = d * np.log(np.pi) / 2 - gammaln(d / 2 + 1)
    V = np.log(nn) * d + const
    Vdr = np.log(d) + (d - 1) * np.log(nn) + const
    mle = -V
    mu = float(np.quantile(mle, 0.01)) - 10
    fit.set_likelihood(V, Vdr, mu)
    t0 = time.time()
    z0 = fit.ridge_init(mle - mu)
    t1 = time.time()
    st = fit.stage_times()
    print(f"ridge_init {t1-t0:.3f}s gram={st['ridge_gram_s']:.3f} solve={st['ridge_solve_s']:.3f}")
    fit.objective(z0)
    t0 = time.time()
    for _ in range(evals):
        fit.objective(z0)
    t1 = time.time()
    st = fit.stage_times()
    per = st["objective_kernel_s"] / st["objective_launches"]
    print(f"objective wall/eval {(t1-t0)/evals*1e3:.3f} ms ; kernel {per*1e3:.3f} ms ; "
          f"{st['objective_bytes_per_launch']/per/1e12:.3f} TB/s")
    t0 = time.time(); f = fit.transform(z0, mu); t1 = time.time()
    print(f"transform {t1-t0:.3f}s")
    w = fit.weights_cholesky(z0)
    t0 = time.time(); p = ctx.predict_mean(c.lower(d), xd, xu, w, mu); t1 = time.time()
    print(f"predict {t1-t0:.3f}s  {n/(t1-t0):.3e} cells/s  max|pred-f|={np.abs(p-f).max():.2e}")
    flops_trsm = n * m * m
    print(f"TRSM {flops_trsm/st['trsm_s']/1e12:.2f} TF/s ; Gram {flops_trsm/st['ridge_gram_s']/1e12:.2f} TF/s")

if __name__ == "__main__":
    a = [int(float(v)) for v in sys.argv[1:4]]
    main(*a, *(sys.argv[4:5]))
