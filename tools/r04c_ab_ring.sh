#!/bin/bash
# A/B on one box: default / ring for every qualifying all-quadrant launch (RING=2) / the 3-of-4 rule for the last round (KEEP8=6)
B="python bench.py --landmark-method device --cpu-sample 0 --steps 6 --warmup 2 --extra-steps 0"
O=gpurun_out/ab; mkdir -p $O; rm -f $O/*
$B > $O/warm.json 2> $O/warm.err
for rep in 1 2 3; do
  $B > $O/default_$rep.json 2> $O/default_$rep.err
  MELLON_AMD_EXPERIMENTAL=1 MELLON_AMD_GEMM_RING=2 $B > $O/ring2_$rep.json 2> $O/ring2_$rep.err
  MELLON_AMD_EXPERIMENTAL=1 MELLON_AMD_GEMM_KEEP8=6 $B > $O/keep6_$rep.json 2> $O/keep6_$rep.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        st = d["stages_s"]
        print(f"{f:40s} {d['ms_per_step']:8.2f} ms  chol {st['cholesky_s']*1e3:5.2f} gram {st['ridge_gram_s']*1e3:5.2f} solve {st['ridge_solve_s']*1e3:5.2f} rebuild {st['precond_rebuild_s']*1e3:5.2f} obj {st['objective_kernel_s']*1e3:6.2f} sub {st['objective_sub_kernel_s']*1e3:5.2f} evals {d['config']['objective_evaluations']}")
    except Exception as e:
        print(f, "ERR", e)
PY
