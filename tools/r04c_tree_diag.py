"""The guard-off stress variant of tests/test_gpu_round4.py::test_inputs_the_shortcuts_were_not_tuned_on[tree] on its own, with the
solver's pause trace: which path does 'mixed, rebuild never declines' take under the GEMM policy of the environment?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MELLON_AMD_EXPERIMENTAL"] = "1"
os.environ["MELLON_AMD_REBUILD_RANGE"] = "1e300"
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mellon_amd
from mellon_amd import _lib
from test_gpu_round4 import tree_cells
ctx = _lib.default_context()
n, d, m = 1_000_000, 20, 2000
rng = np.random.default_rng(11)
x = tree_cells(n, d, rng)
xd = ctx.to_device(np.ascontiguousarray(x))
nn = ctx.nn_distances(xd, xd)
lm = ctx.kmeans(x[:100_000], m, seed=42)
if "--suite-order" in sys.argv:        # the fits tests/test_gpu_round4.py runs before the guard-off ones, in its order
    for env in ({"MELLON_AMD_MIXED": "0", "MELLON_AMD_SUBSAMPLE": "0", "MELLON_AMD_REBUILD": "0"}, {"MELLON_AMD_MIXED": "0"}, {}):
        for k in ("MELLON_AMD_MIXED", "MELLON_AMD_SUBSAMPLE", "MELLON_AMD_REBUILD", "MELLON_AMD_REBUILD_RANGE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
        est.fit_predict(xd)
        print(f"(before) {env} success={est.opt_state.success} nfev={est.opt_state.nfev}", flush=True)
        est._fit.close()
    for k in ("MELLON_AMD_MIXED", "MELLON_AMD_SUBSAMPLE", "MELLON_AMD_REBUILD"):
        os.environ.pop(k, None)
    os.environ["MELLON_AMD_REBUILD_RANGE"] = "1e300"
for mixed in ("0", "1"):
    os.environ["MELLON_AMD_MIXED"] = mixed
    est = mellon_amd.DensityEstimator(landmarks=lm, nn_distances=nn, check_rank=False)
    dens = est.fit_predict(xd)
    st = est._fit.stage_times()
    z = np.asarray(est.pre_transformation)
    loss, grad = est._fit.objective(z)
    print(f"mixed={mixed} success={est.opt_state.success} nfev={est.opt_state.nfev} loss={loss:.12e} gmax={np.abs(grad).max():.3e} "
          f"rebuilds={st.get('n_rebuild')} reverts={st.get('n_revert')} skipped={st.get('n_rebuild_skipped')}", flush=True)
    est._fit.close()
