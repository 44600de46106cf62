#!/bin/bash
# k-means with pruned / queued sweeps against the unpruned sweeps: same centres, timing, kernel breakdown
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_km; mkdir -p $O
cat > $O/probe.py <<'PY'
import sys, time, os
sys.path.insert(0, ".")
import numpy as np
import bench
from mellon_amd import _lib
ctx = _lib.default_context()
for d in (20, 50):
    x = bench.gaussian_mixture(1_000_000, d, 3); xd = ctx.to_device(x)
    for rep in range(2):
        t0 = time.perf_counter(); c, it, inertia = ctx.kmeans(xd, 5000, seed=42, return_info=True)
        print("kmeans d", d, "prune", os.environ.get("MELLON_AMD_KM_PRUNE", "1"), round(time.perf_counter() - t0, 3), "s", it, "sweeps", repr(inertia), flush=True)
    np.save(sys.argv[1] + f"_{d}.npy", c)
PY
export MELLON_AMD_EXPERIMENTAL=1
MELLON_AMD_KM_PRUNE=0 timeout 600 python $O/probe.py $O/c_plain > $O/log.txt 2>&1
MELLON_AMD_KM_PRUNE=1 timeout 600 python $O/probe.py $O/c_prune >> $O/log.txt 2>&1
python - >> $O/log.txt <<PY
import numpy as np
for d in (20, 50):
    a = np.load("$O/c_plain_%d.npy" % d); b = np.load("$O/c_prune_%d.npy" % d)
    print("d", d, "centres pruned vs plain: max abs diff", np.abs(a - b).max(), "scale", np.abs(a).max())
PY
cat $O/log.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "kmeans or landmarks or labels" > $O/tests_km.log 2>&1 < /dev/null; tail -5 $O/tests_km.log
bash tools/r06_km_trace.sh | head -16
