#!/bin/bash
# same-box A/B of the C3 step: each argument is an environment ("A=1 B=2"); interleaved, REPS times; BENCH_ARGS adds bench flags
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${REPS:-2}); do
  for e in "$@"; do
    env MELLON_AMD_EXPERIMENTAL=1 $e python bench.py --landmark-method device --steps ${STEPS:-5} --warmup 2 --cpu-sample 0 --extra-steps 0 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('[$e] $BENCH_ARGS', round(d['ms_per_step'],2), 'ms  pass', round(d['roofline']['avg_launch_ms'],3), 'evals', d['config']['objective_evaluations'], 'passes', round(d['config']['objective_full_pass_equivalents'],2), 'rebuild_s', d['stages_s']['precond_rebuild_s'])"
  done
done
