#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_km; mkdir -p $O
export MELLON_AMD_EXPERIMENTAL=1
cat > $O/p.py <<'PY'
import sys, time, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_gpu_round5 import _tree
import bench
from mellon_amd import _lib
ctx = _lib.default_context()
what = sys.argv[1]; n = int(sys.argv[2])
x = _tree(n, 20, 15) if what == "tree" else bench.gaussian_mixture(n, 20, 3)
xd = ctx.to_device(x)
for rep in range(int(sys.argv[3])):
    t0 = time.perf_counter(); c, it, inertia = ctx.kmeans(xd, 5000, seed=42, return_info=True); print("kmeans", what, n, "prune", os.environ.get("MELLON_AMD_KM_PRUNE"), round(time.perf_counter() - t0, 3), "s", it, "sweeps", inertia, flush=True)
PY
MELLON_AMD_KM_DEBUG=2 timeout 600 python $O/p.py mix 300000 1 > $O/verify_mix.txt 2>&1
grep -c verify $O/verify_mix.txt; grep verify $O/verify_mix.txt | sort | uniq -c | sort -rn | head -8; grep "kmeans\|rror" $O/verify_mix.txt | head
for W in mix tree; do for P in 1 0; do
MELLON_AMD_KM_PRUNE=$P MELLON_AMD_KM_DEBUG=1 timeout 300 python $O/p.py $W 1000000 2 > $O/debug_${W}_$P.txt 2>&1
awk 'NR%6==1' $O/debug_${W}_$P.txt | grep sweep | head -12; grep "kmeans\|rror" $O/debug_${W}_$P.txt
done; done
