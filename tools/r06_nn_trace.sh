#!/bin/bash
# kernel-level breakdown of the (pruned) exact 1-NN search at the C3 shape
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_nn_trace; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O/db -o nn -- python tools/nn_probe_c3.py > $O/log.txt 2>&1
DB=$(find $O/db -name "*.db" | head -1)
python - $DB > $O/nn_kernels.txt <<'PY'
import sqlite3, sys, re
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
# the second call: after the last k_prune_sample launch
idx = [i for i, r in enumerate(rows) if "k_prune_sample" in r[0]]
rows = rows[idx[-1]:] if idx else rows
t0, t1 = rows[0][1], rows[-1][2]
from collections import defaultdict
tot = defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)[:70]
    tot[n][0] += 1; tot[n][1] += (e - s) / 1e6
print(f"span {(t1 - t0) / 1e6:.1f} ms, kernels busy {sum(v[1] for v in tot.values()):.1f} ms, {len(rows)} launches")
for n, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{n:72s} {c:6d} {ms:9.3f} ms")
PY
rm -rf $O/db; cat $O/nn_kernels.txt; tail -2 $O/log.txt
