#!/bin/bash
# kernel trace + stats of one BASELINE config's bench line: summary table (rocprofv3 --stats) and the last step's timeline
# usage: bash tools/r06_cfg_trace.sh c2|c4|c5
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
C=$1; O=gpurun_out/r06_cfg_$C; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats -d $O/db -o t -- python bench.py --config $C --steps 4 --warmup 2 --cpu-sample 0 --extra-steps 0 > $O/bench_under_rocprof.json 2> $O/bench.err
DB=$(find $O/db -name "*.db" | head -1)
python tools/config_timeline.py $DB 600 > $O/timeline.txt 2> $O/tl.err
python - $DB > $O/summary.txt <<'PY'
import sqlite3, sys, re
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, count(*), avg(end-start), sum(end-start), min(end-start), max(end-start) from kernels group by name order by sum(end-start) desc").fetchall()
tot = sum(r[3] for r in rows)
print(f"# rocprofv3 --kernel-trace --stats -- python bench.py --config ... --steps 4 --warmup 2 (set-up, warm-up and timed steps together)")
print(f"# {len(rows)} kernels, {tot / 1e6:.1f} ms of kernel time in total")
print(f"{'kernel':100s} {'calls':>7s} {'avg us':>10s} {'total ms':>10s} {'%':>6s}")
for n, c, a, s, mn, mx in rows[:45]:
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)
    print(f"{n[:100]:100s} {c:7d} {a / 1e3:10.2f} {s / 1e6:10.2f} {100 * s / tot:6.2f}")
PY
find $O -name "*.db" -delete; rm -rf $O/db
tail -45 $O/timeline.txt; cut -c1-300 $O/bench_under_rocprof.json
