#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_d; mkdir -p $O
timeout 300 python tools/cprofile_c2.py > $O/cprofile_c2.txt 2>&1; head -60 $O/cprofile_c2.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "rebuild or hard_data or golden or sharded or c2_full" > $O/tests.log 2>&1 < /dev/null; tail -3 $O/tests.log
timeout 600 python bench.py --steps 8 --warmup 2 --cpu-sample 0 --extra-steps 0 > $O/bench.json 2> $O/bench.err < /dev/null; cut -c1-300 $O/bench.json
timeout 300 python bench.py --config c2 --steps 20 --warmup 3 --cpu-sample 0 --extra-steps 0 > $O/bench_c2.json 2> $O/bench_c2.err < /dev/null; cut -c1-300 $O/bench_c2.json
