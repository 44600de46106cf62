cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_round4.py tests/test_gpu_estimators.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1
for i in 1 2; do timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --extra-steps 4 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],2), round(j['ms_per_step_host_to_host'],2), j['stages_s']['kernel_matrix_s'], round(j['roofline']['avg_launch_ms'],3), j['host_s'])"; done
