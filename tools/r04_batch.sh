cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/ -q -m gpu > gpurun_out/r04_tests_d.log 2>&1; grep -n "passed\|failed\|Error" gpurun_out/r04_tests_d.log | tail -8
for mode in "default" "MELLON_AMD_UPLOAD_PIPELINE=0"; do
  echo "== $mode"
  env $( [ "$mode" = default ] || echo $mode ) python bench.py --cpu-sample 0 --landmark-method device --steps 4 --warmup 1 --extra-steps 3 > gpurun_out/r04_bench_d.json 2> gpurun_out/r04_bench_d.err || tail -5 gpurun_out/r04_bench_d.err
  python -c "
import json;d=json.load(open('gpurun_out/r04_bench_d.json'));print({k:round(d[k],2) for k in ('ms_per_step','ms_per_step_host_to_host','ms_per_step_mixed')}, d['config']['objective_evaluations']); print(d['stages_s'])"
done
