#!/bin/bash
# Round-6 measurement batch (run through gpurun): rocprofv3 summary + PMC passes of the default bench command, the default bench
# line (sklearn landmarks, CPU baselines), C2 / C4 / C5 lines and their kernel traces, the emulated-rank table, one step's
# kernels / timeline / gaps, the drop-in profile, the robustness sweep.  Everything under gpurun_out/r06/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
COMMIT=$1 bash tools/profile_round.sh r06 > $O/profile_round.log 2>&1 < /dev/null
timeout 900 python bench.py > $O/r06_bench_c3_1gpu.json 2> $O/bench_c3.err < /dev/null
for c in c2 c4 c5; do
  timeout 600 python bench.py --config $c > $O/r06_bench_$c.json 2> $O/bench_$c.err < /dev/null
  bash tools/r06_cfg_trace.sh $c > $O/cfg_$c.log 2>&1 < /dev/null
done
timeout 600 python tools/emulate_rank.py 1 2 4 8 > $O/r06_emulated_ranks.json 2> $O/emu.err < /dev/null
timeout 300 python tools/dropin_profile.py > $O/r06_dropin_profile.txt 2> $O/dropin.err < /dev/null
mkdir -p $O/trace
timeout 300 rocprofv3 --kernel-trace -d $O/trace/db -o one -- python tools/one_step.py > $O/trace/one.log 2>&1 < /dev/null
DB=$(find $O/trace/db -name "*.db" | head -1)
if [ -n "$DB" ]; then
  python tools/step_timeline.py $DB > $O/r06_step_timeline.txt 2> $O/trace/tl.err
  python tools/step_kernels.py $DB > $O/r06_step_kernels.txt 2> $O/trace/sk.err
  python tools/gap_report.py $DB > $O/r06_step_gaps.txt 2> $O/trace/gap.err
fi
rm -rf $O/trace/db
bash tools/r06_nn_trace.sh > $O/r06_nn_trace.txt 2>&1 < /dev/null
KM_D=20 bash tools/r06_km_trace.sh > $O/r06_km_trace.txt 2>&1 < /dev/null
bash tools/r06_km.sh > $O/r06_kmeans_ab.txt 2>&1 < /dev/null
timeout 1500 python tools/robustness_sweep_large.py > $O/r06_robustness_sweep_large.txt 2> $O/robust.err < /dev/null
for c in c3_1gpu c2 c4 c5; do python - $O/r06_bench_$c.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], round(d["ms_per_step"], 2), "ms", round(d["value"]), d["unit"], "roofline", round(d["roofline"]["frac"], 3),
          "cpu", (d.get("cpu_baseline") or {}).get("value"), "h2h", d.get("ms_per_step_host_to_host"), "predict", (d.get("predict") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
head -12 $O/r06_step_kernels.txt; head -8 $O/r06_dropin_profile.txt; tail -12 $O/r06_robustness_sweep_large.txt
