"""Are the inputs of tests/test_gpu_round4.py::test_inputs_the_shortcuts_were_not_tuned_on the same bits from run to run?
Prints checksums of the device k-means landmarks, the exact 1-NN distances and the default fit's log-density (twice)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import mellon_amd
from mellon_amd import _lib
from test_gpu_round4 import tree_cells
h = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
ctx = _lib.default_context()
n, d, m = 1_000_000, 20, 2000
x = tree_cells(n, d, np.random.default_rng(11))
xd = ctx.to_device(np.ascontiguousarray(x))
nn = ctx.nn_distances(xd, xd)
print("nn", h(np.asarray(nn.to_host() if hasattr(nn, "to_host") else nn)))
for rep in range(2):
    lm = ctx.kmeans(x[:100_000], m, seed=42)
    print("landmarks", rep, h(lm))
os.environ["MELLON_AMD_MIXED"] = "0"
for rep in range(2):
    est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
    dens = est.fit_predict(xd)
    print("default fp64 fit", rep, "nfev", est.opt_state.nfev, "dens", h(dens))
    est._fit.close()
