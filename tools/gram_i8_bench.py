"""The preconditioner's integer Gram in isolation at the C3 shape (60 000 sampled cells x 5000 landmarks): ms per call of
digit extraction + int8 GEMM + sum of the k-chunks, against the fp64 GEMM it replaces."""
import os as _os; _os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")   # this tool turns experiment knobs (csrc/mln_options.h)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mellon_amd import _lib
ctx = _lib.default_context()
rng = np.random.default_rng(0)
a = rng.random((60000, 5000))
got, ms = ctx.diag_gram_i8(a, reps=5)
print("int8 gram ms", ms, "Pop/s (9 digit products, lower tiles)", 9 * 60000 * 5000 * 5000 / ms / 1e12)
if "--fp64" in sys.argv:
    print("dgemm lower ms", ctx.diag_dgemm(1, 0, 5000, 5000, 60000, lower_only=1, split_k=7, reps=3))
