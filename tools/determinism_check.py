"""Is fit_predict bit-reproducible on identical inputs?  (same process, and across processes via a checksum)"""
import hashlib
import sys

import numpy as np

sys.path.insert(0, ".")
import mellon_amd
from mellon_amd import _lib
import bench

n, d, m = 200_000, 50, 2000
x = bench.gaussian_mixture(n, d, 3)
from threadpoolctl import threadpool_limits
with threadpool_limits(limits=1):
    lm = bench.make_landmarks(x, m)
ctx = _lib.default_context()
xd = ctx.to_device(x)
nn = ctx.nn_distances(xd, xd, self_offset=0)
outs = []
for rep in range(3):
    est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn)
    dens = est.fit_predict(xd)
    outs.append((est.loss_func.n_eval, hashlib.sha1(np.ascontiguousarray(dens).tobytes()).hexdigest()[:12],
                 hashlib.sha1(np.ascontiguousarray(est.pre_transformation).tobytes()).hexdigest()[:12]))
    est._fit.close()
print(outs)
print("landmarks", hashlib.sha1(lm.tobytes()).hexdigest()[:12], "nn", hashlib.sha1(nn.tobytes()).hexdigest()[:12])
