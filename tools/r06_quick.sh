#!/bin/bash
# quick step-level check: C3 / C2 / C4 bench lines without CPU baselines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_quick; mkdir -p $O
for c in c3 c2 c4; do
  timeout 600 python bench.py --config $c --steps 10 --warmup 3 --cpu-sample 0 --extra-steps 0 > $O/bench_$c.json 2> $O/bench_$c.err < /dev/null
  python - $O/bench_$c.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(sys.argv[1], round(d["ms_per_step"], 2), "ms, pass", round(d["roofline"].get("avg_launch_ms", 0), 3), d.get("host_s"))
PY
done
