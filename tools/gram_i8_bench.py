import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from mellon_amd import _lib
ctx = _lib.default_context()
rng = np.random.default_rng(0)
a = rng.random((60000, 5000))
got, ms = ctx.diag_gram_i8(a, reps=5)
print("int8 gram ms", ms, "Tops", 6 * 60000 * 5000 * 5000 / ms / 1e9)
print("dgemm lower ms", ctx.diag_dgemm(1, 0, 5000, 5000, 60000, lower_only=1, split_k=7, reps=3))
