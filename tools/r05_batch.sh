#!/bin/bash
# Round-5 measurement batch (run through gpurun): the default bench line (sklearn landmarks, CPU baseline), C2 / C4 / C5 with their
# CPU baselines, the emulated-rank table, one step's kernels / timeline, the drop-in profile.  Everything under gpurun_out/r05/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 900 python bench.py > $O/r05_bench_c3_1gpu.json 2> $O/bench_c3.err < /dev/null
timeout 600 python bench.py --config c2 > $O/r05_bench_c2.json 2> $O/bench_c2.err < /dev/null
timeout 600 python bench.py --config c4 > $O/r05_bench_c4.json 2> $O/bench_c4.err < /dev/null
timeout 600 python bench.py --config c5 > $O/r05_bench_c5.json 2> $O/bench_c5.err < /dev/null
timeout 600 python tools/emulate_rank.py 1 2 4 8 > $O/r05_emulated_ranks.json 2> $O/emu.err < /dev/null
timeout 300 python tools/dropin_profile.py > $O/r05_dropin_profile.txt 2> $O/dropin.err < /dev/null
mkdir -p $O/trace
timeout 300 rocprofv3 --kernel-trace -d $O/trace/db -o one -- python tools/one_step.py > $O/trace/one.log 2>&1 < /dev/null
DB=$(find $O/trace/db -name "*.db" | head -1)
if [ -n "$DB" ]; then
  python tools/step_timeline.py $DB > $O/r05_step_timeline.txt 2> $O/trace/tl.err
  python tools/step_kernels.py $DB > $O/r05_step_kernels.txt 2> $O/trace/sk.err
  python tools/gap_report.py $DB > $O/r05_step_gaps.txt 2> $O/trace/gap.err
fi
rm -rf $O/trace/db
for c in c3_1gpu c2 c4 c5; do python - $O/r05_bench_$c.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], round(d["ms_per_step"], 2), "ms", round(d["value"]), d["unit"], "roofline", round(d["roofline"]["frac"], 3),
          "cpu", (d.get("cpu_baseline") or {}).get("value"), "h2h", d.get("ms_per_step_host_to_host"), d.get("ms_per_step_host_to_host_pageable"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
head -12 $O/r05_step_kernels.txt; head -8 $O/r05_dropin_profile.txt
