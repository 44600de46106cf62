#!/bin/bash
# kernel-level breakdown of mln_kmeans at the C3 shape (1e6 x KM_D (default 20) -> 5000 centres)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_km_trace; mkdir -p $O
cat > $O/probe.py <<'PY'
import sys, time
sys.path.insert(0, ".")
import bench
from mellon_amd import _lib
ctx = _lib.default_context()
import os
D = int(os.environ.get("KM_D", "20"))
x = bench.gaussian_mixture(1_000_000, D, 3); xd = ctx.to_device(x)
for rep in range(2):
    t0 = time.perf_counter(); c = ctx.kmeans(xd, 5000, seed=42); print("kmeans", round(time.perf_counter() - t0, 3), "s", flush=True)
PY
timeout 600 rocprofv3 --kernel-trace -d $O/db -o km -- python $O/probe.py > $O/log.txt 2>&1
DB=$(find $O/db -name "*.db" | head -1)
python - $DB > $O/km_kernels.txt <<'PY'
import sqlite3, sys, re
from collections import defaultdict
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
# the second call: every level of a call starts with k_km_colmax
idx = [i for i, r in enumerate(rows) if "k_km_colmax" in r[0]]
rows = rows[idx[len(idx) // 2]:]
t0, t1 = rows[0][1], rows[-1][2]
tot = defaultdict(lambda: [0, 0.0]); idle = 0.0; prev = rows[0][1]
for n, s, e in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)[:70]
    tot[n][0] += 1; tot[n][1] += (e - s) / 1e6; idle += max(s - prev, 0) / 1e6; prev = e
big = sorted(((rows[i + 1][1] - rows[i][2]) / 1e6, i) for i in range(len(rows) - 1))[-8:]
for g, i in sorted(big, key=lambda t: t[1]):
    print(f"gap {g:7.2f} ms at {(rows[i][2] - t0) / 1e6:8.1f} ms  after {rows[i][0][:40]}  before {rows[i + 1][0][:40]}")
print(f"span {(t1 - t0) / 1e6:.1f} ms, kernels busy {sum(v[1] for v in tot.values()):.1f} ms, idle between launches {idle:.1f} ms, {len(rows)} launches")
for n, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{n:72s} {c:6d} {ms:9.3f} ms")
PY
rm -rf $O/db; cat $O/km_kernels.txt; grep kmeans $O/log.txt
