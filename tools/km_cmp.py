import os, sys
sys.path.insert(0, "/root/repo")
os.environ["MELLON_AMD_EXPERIMENTAL"] = "1"
os.environ["MELLON_AMD_KM_LEVELS"] = "1"
import numpy as np
from oracle import mellon_oracle as mo
from mellon_amd import _lib
ctx = _lib.default_context()
x = mo.gaussian_mixture(260_000, 12, seed=5)
for seed in (11, 12, 13):
    cb, itb, inb = ctx.kmeans(x, 400, seed=seed, return_info=True)
    os.environ["MELLON_AMD_KM_BOUNDS"] = "0"
    cp, itp, inp = ctx.kmeans(x, 400, seed=seed, return_info=True)
    del os.environ["MELLON_AMD_KM_BOUNDS"]
    print("w64", os.environ.get("MELLON_AMD_ROWMIN_W64", "1"), "seed", seed, "bounds", itb, inb, "plain", itp, inp, flush=True)
