#!/bin/bash
# Round-4c evidence batch on the GPU box (through gpurun): tests, profile round, configs, emulated ranks, full bench, trace.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.txt 2>&1; grep -E "passed|failed" gpurun_out/pytest_gpu.txt | tail -2
COMMIT=$1 bash tools/profile_round.sh r04 > gpurun_out/profile_round.log 2>&1; tail -3 gpurun_out/profile_round.log
python tools/run_configs.py > gpurun_out/r04_configs_c2_c4_c5.json 2> gpurun_out/configs.err; tail -c 400 gpurun_out/r04_configs_c2_c4_c5.json
python tools/emulate_rank.py 1 2 4 8 > gpurun_out/r04_emulated_ranks.json 2> gpurun_out/emulated.err
python -c "
import json;d=json.load(open('gpurun_out/r04_emulated_ranks.json'))
for k,v in d.items(): print(k, {a:round(b,1) if isinstance(b,float) else b for a,b in v.items() if a in ('step_ms','evaluations','objective_kernels_ms','kernel_matrix_ms','chol_Lp_ms','gram_and_solves_ms','chol_C_inverses_ms','sub_passes_ms','rebuild_ms','rebuilds','speedup_without_communication','composed_global_trajectory_ms')})"
bash tools/r04c_trace.sh final > gpurun_out/trace_final.txt 2>&1; head -14 gpurun_out/trace_final.txt
for c in c2 c4 c5; do python bench.py --config $c --cpu-sample 0 > gpurun_out/r04_bench_$c.json 2> gpurun_out/bench_$c.err; done
python bench.py > gpurun_out/r04_bench_c3_1gpu.json 2> gpurun_out/bench_c3.err; tail -c 300 gpurun_out/r04_bench_c3_1gpu.json
