cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/ -q -m gpu > gpurun_out/tests_gpu.log 2>&1; tail -3 gpurun_out/tests_gpu.log
python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; tail -c 600 gpurun_out/bench_full.json
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1e -o bench -- python bench.py --cpu-sample 0 --steps 2 --warmup 1 > gpurun_out/prof_r1e.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc5_fetch -o bench -- python bench.py --cpu-sample 0 --steps 1 --warmup 0 > gpurun_out/pmc5_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc5_write -o bench -- python bench.py --cpu-sample 0 --steps 1 --warmup 0 > gpurun_out/pmc5_write.log 2>&1
for d in prof_r1e pmc5_fetch pmc5_write; do f=$(find gpurun_out/$d -name "*.db" | head -1); echo $d $f; done
python tools/rocpd_summary.py $(find gpurun_out/prof_r1e -name "*.db" | head -1) $(find gpurun_out/pmc5_fetch -name "*.db" | head -1) $(find gpurun_out/pmc5_write -name "*.db" | head -1) > gpurun_out/r01e_summary.txt 2>&1
python tools/rocpd_timeline.py $(find gpurun_out/prof_r1e -name "*.db" | head -1) > gpurun_out/r01e_timeline.txt 2>&1
head -30 gpurun_out/r01e_summary.txt
find gpurun_out -name "*.db" -size +30M -delete
