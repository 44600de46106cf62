import os as _os; _os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")   # this tool turns experiment knobs (csrc/mln_options.h)
import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["MELLON_AMD_MIXED"]="1"; os.environ["MELLON_AMD_MIXED_FTOL"]=sys.argv[1]
import numpy as np, bench, mellon_amd
from mellon_amd import _lib
ctx=_lib.default_context()
n,d,m=1_000_000,50,5000
x=bench.gaussian_mixture(n,d,3); lm=bench.make_landmarks(x,m); xd=ctx.to_device(x); nn=ctx.nn_distances(xd)
est=mellon_amd.DensityEstimator(landmarks=lm,nn_distances=nn); est.prepare_inference(xd)
os.environ["MELLON_AMD_TRACE"]="2"
z,l,ne,ni,st=est._fit.map_solve(est.initial_value)
print("evals",ne,"iters",ni)
