"""Accuracy of the persistent-row kernels' element epilogue (cov_epilogue.h) on the device: kernel matrix and fused
predictive mean against a long-double evaluation of the reference's formulas (util.py:351-366, cov.py k()).
    python tools/epilogue_accuracy.py          (on a GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mellon_amd
from mellon_amd import _lib

LD = np.longdouble
rng = np.random.default_rng(0)
n, m, d = 8192, 512, 50
x = rng.normal(size=(n, d)) * 1.3
y = x[rng.choice(n, m, replace=False)] + 0.05 * rng.normal(size=(m, d))
y[:8] = x[:8]                                            # coincident pairs
xl, yl = x.astype(LD), y.astype(LD)
sq = (xl * xl).sum(1)[:, None] - 2 * (xl @ yl.T) + (yl * yl).sum(1)[None, :] + LD(1e-12)
dist = np.sqrt(np.maximum(sq, 0))
ls = 9.0
forms = {"Matern52": lambda r: (1 + np.sqrt(LD(5)) * r + 5 * r * r / 3) * np.exp(-np.sqrt(LD(5)) * r),
         "Matern32": lambda r: (1 + np.sqrt(LD(3)) * r) * np.exp(-np.sqrt(LD(3)) * r),
         "ExpQuad": lambda r: np.exp(-r * r / 2), "Exponential": lambda r: np.exp(-r / 2)}
ctx = _lib.default_context()
for name, f in forms.items():
    want = f(dist / LD(ls))
    cov = getattr(mellon_amd.cov, name)(ls)
    got = np.asarray(cov(x, y))
    rel = np.abs((got.astype(LD) - want) / want)
    a = np.abs(got.astype(LD) - want)
    w = rng.normal(size=m)
    pm = ctx.predict_mean(cov.lower(d), x, y, w, 0.25)
    pw = (want @ w.astype(LD) + LD(0.25)).astype(np.float64)
    print(f"{name:12s} K: max rel err {float(rel.max()):.2e} (mean {float(rel.mean()):.2e}), max abs err {float(a.max()):.2e}, "
          f"min K {float(want.min()):.2e};  predict_mean max abs err {np.abs(pm - pw).max():.2e} (|mean| ~ {np.abs(pw).mean():.2f})")
