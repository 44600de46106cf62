import os as _os; _os.environ.setdefault("MELLON_AMD_EXPERIMENTAL", "1")   # this tool turns experiment knobs (csrc/mln_options.h)
import sys; sys.path.insert(0,"/root/repo")
import numpy as np, mellon_amd
from oracle import mellon_oracle as mo
from mellon_amd import _lib
g=np.load("/root/repo/tests/golden/c3_sub_density.npz")
n,d,m,keep=int(g["n"]),int(g["dims"]),int(g["m"]),int(g["keep_every"])
x=mo.gaussian_mixture(n,d,seed=int(g["seed"])); idx=np.sort(np.random.default_rng(int(g["landmark_seed"])).choice(n,m,replace=False))
nn=_lib.default_context().nn_distances(x)
for mixed in ("1","0"):
    import os; os.environ["MELLON_AMD_MIXED"]=mixed
    est=mellon_amd.DensityEstimator(landmarks=x[idx],nn_distances=nn); dens=est.fit_predict(x)
    ref=g["log_density_sub"]; a=dens[::keep]
    print("mixed",mixed,"rel_max",np.abs(a-ref).max()/np.abs(ref).max(),"rel_std",np.std(a-ref)/np.std(ref),"evals",est.loss_func.n_eval, "z rel", np.abs(est.pre_transformation-g["pre_transformation"]).max()/np.abs(g["pre_transformation"]).max())
