#!/bin/bash
# Round profile on the GPU box (run through gpurun): rocprofv3 kernel trace + stats of the default bench command, then the
# PMC passes for HBM traffic (FETCH_SIZE, WRITE_SIZE in SEPARATE runs, --kernel-trace only) and MFMA utilisation.
# Results land in gpurun_out/prof_$TAG/; tools/make_profiles.py turns the databases into profiles/*.json, and the text
# summaries are copied to profiles/ by hand.   usage: bash tools/profile_round.sh r03 [bench args...]
set -u
TAG=${1:-r03}; shift || true
ARGS=${@:---steps 2 --warmup 1 --cpu-sample 0 --landmark-method device}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprof flags...
  local name=$1; shift
  timeout 900 rocprofv3 "$@" -d "$OUT/$name" -o "$name" -- python "$ROOT/bench.py" $ARGS > "$OUT/$name.bench.json" 2> "$OUT/$name.err"
  echo "$name rc=$?"
}
run stats --kernel-trace --stats
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
run sq --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
# the databases are too large to travel back: reduce them here, keep the summaries
COMMIT=${COMMIT:-unrecorded}
SHA=$(sha256sum "$ROOT/mellon_amd/csrc/objective.hip" | cut -c1-16)
cd "$OUT" && mkdir -p profiles
python - "$OUT" "$TAG" <<'PY' > "$OUT/${TAG}_bench_c3_summary.txt" 2> "$OUT/summary.err"
import sqlite3, sys, re, json
out, tag = sys.argv[1], sys.argv[2]
def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name)[:120]
con = sqlite3.connect(f"{out}/stats/stats_results.db")
rows = con.execute("select name, count(*), avg(duration), sum(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[3] for r in rows)
print(f"# rocprofv3 --kernel-trace --stats -- python bench.py (see {tag}_bench_c3_1gpu_under_rocprof.json for the line this run printed)")
print(f"# {len(rows)} kernels, {tot / 1e6:.1f} ms of kernel time in total")
print(f"{'kernel':122s} {'calls':>7s} {'avg us':>12s} {'total ms':>10s} {'%':>6s} {'min us':>10s} {'max us':>10s}")
for n, c, a, s, mn, mx in rows:
    print(f"{short(n):122s} {c:7d} {a / 1e3:12.2f} {s / 1e6:10.2f} {100 * s / tot:6.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f}")
# The objective kernels are launched for every evaluation of the device-resident solver and gated on its state: a row above
# mixes launches that streamed the whole buffer, launches over the row subsample and no-ops.  Per kernel: the launches within
# 2x of the longest one ("full size") on their own.
print()
print("# objective kernels, FULL-SIZE launches only (duration > half of the kernel's longest launch)")
print(f"{'kernel':122s} {'calls':>7s} {'avg us':>12s} {'total ms':>10s} {'min us':>10s} {'max us':>10s}")
for (n,) in con.execute("select distinct name from kernels where name like '%k_objective%'").fetchall():
    d = [r[0] for r in con.execute("select duration from kernels where name = ?", (n,)).fetchall()]
    full = [v for v in d if v > 0.5 * max(d)]
    if max(d) < 2e5:
        continue
    print(f"{short(n):122s} {len(full):7d} {sum(full) / len(full) / 1e3:12.2f} {sum(full) / 1e6:10.2f} {min(full) / 1e3:10.2f} {max(full) / 1e3:10.2f}")
PY
python "$ROOT/tools/make_profiles.py" "$OUT/stats/stats_results.db" "$OUT/fetch/fetch_results.db" "$OUT/write/write_results.db" "$OUT/sq/sq_results.db" 1000000 5000 5008 "$TAG" > "$OUT/make_profiles.out" 2> "$OUT/make_profiles.err"
python - "$OUT/profiles/objective_traffic.json" "$COMMIT" "$SHA" <<'PY'
import json, sys
p, commit, sha = sys.argv[1:4]
try:
    d = json.load(open(p)); d["measured_at_commit"] = commit; d["objective_hip_sha16"] = sha
    json.dump(d, open(p, "w"), indent=1)
except Exception as e:
    print("stamp failed:", e)
PY
python "$ROOT/tools/pmc_summary.py" "$OUT/sq/sq_results.db" > "$OUT/${TAG}_pmc_sq.txt" 2>> "$OUT/summary.err"
cp "$OUT/stats.bench.json" "$OUT/${TAG}_bench_c3_1gpu_under_rocprof.json"
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write" "$OUT/sq"
ls -la "$OUT" "$OUT/profiles"
