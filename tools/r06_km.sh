#!/bin/bash
# k-means (1e6 cells -> 5000 centres) with group bounds against the Hamerly-bound sweeps of the same build: isotropic
# mixture (bench.py's cells, d = 20 and 50) and the diffusion-map-like tree of the robustness sweeps; prints time, sweeps, inertia
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_km; mkdir -p $O
cat > $O/probe.py <<'PY'
import sys, time, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from test_gpu_round5 import _tree
from mellon_amd import _lib
ctx = _lib.default_context()
for what, d in (("mixture", 20), ("mixture", 50), ("tree", 20)):
    x = bench.gaussian_mixture(1_000_000, d, 3) if what == "mixture" else _tree(1_000_000, d, 15)
    xd = ctx.to_device(x)
    for rep in range(3):
        t0 = time.perf_counter(); c = ctx.kmeans(xd, 5000, seed=42); t = time.perf_counter() - t0
    c, it, inertia = ctx.kmeans(xd, 5000, seed=42, return_info=True)
    print(f"{what:8s} d {d:2d}  group bounds {os.environ.get('MELLON_AMD_KM_PRUNE', '1')}  {t:.3f} s  {it} sweeps (both levels)  inertia {inertia:.6g}", flush=True)
PY
export MELLON_AMD_EXPERIMENTAL=1
MELLON_AMD_KM_PRUNE=0 timeout 600 python $O/probe.py > $O/r06_kmeans_ab.txt 2>&1
MELLON_AMD_KM_PRUNE=1 timeout 600 python $O/probe.py >> $O/r06_kmeans_ab.txt 2>&1
cat $O/r06_kmeans_ab.txt
