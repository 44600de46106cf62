#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_km; mkdir -p $O
export MELLON_AMD_EXPERIMENTAL=1
for P in 1 0; do
MELLON_AMD_KM_PRUNE=$P MELLON_AMD_KM_DEBUG=1 timeout 300 python - > $O/debug_tree_$P.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_gpu_round5 import _tree
from mellon_amd import _lib
ctx = _lib.default_context()
x = _tree(1_000_000, 20, 15); xd = ctx.to_device(x)
for rep in range(2):
    t0 = time.perf_counter(); c, it, inertia = ctx.kmeans(xd, 5000, seed=42, return_info=True); print("kmeans", round(time.perf_counter() - t0, 3), "s", it, "sweeps", inertia, flush=True)
PY
grep -c sweep $O/debug_tree_$P.txt; awk 'NR%8==1' $O/debug_tree_$P.txt | head -24; grep kmeans $O/debug_tree_$P.txt
done
