cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/emulate_rank.py 1 2 4 8 > gpurun_out/r04_emulated_ranks.json 2> gpurun_out/r04_emulated.err; python -c "
import json;d=json.load(open('gpurun_out/r04_emulated_ranks.json'))
for k,v in d.items(): print(k, {a:round(b,1) if isinstance(b,float) else b for a,b in v.items() if a in ('step_ms','evaluations','objective_kernels_ms','kernel_matrix_ms','chol_Lp_ms','gram_and_solves_ms','chol_C_inverses_ms','sub_passes_ms','rebuild_ms','rebuilds','fp64_launches','speedup_without_communication')})"
MELLON_AMD_REBUILD=1 python tools/emulate_rank.py 8 > gpurun_out/r04_emulated_ranks_rebuild8.json 2>> gpurun_out/r04_emulated.err; python -c "
import json;d=json.load(open('gpurun_out/r04_emulated_ranks_rebuild8.json'))
for k,v in d.items(): print('rebuild forced',k, {a:round(b,1) if isinstance(b,float) else b for a,b in v.items() if a in ('step_ms','evaluations','objective_kernels_ms','chol_C_inverses_ms','sub_passes_ms','rebuild_ms','rebuilds','fp64_launches')})"
python tools/robustness_sweep_large.py > gpurun_out/r04_robustness_large.txt 2> gpurun_out/r04_robustness_large.err; cat gpurun_out/r04_robustness_large.txt; tail -3 gpurun_out/r04_robustness_large.err
python -m pytest tests/test_gpu_round3.py tests/test_gpu_estimators.py -q -m gpu -x 2>&1 | tail -3
