#!/bin/bash
# GPU suite (all of it), Gram micro-benchmark (A/B of the LDS store mapping is by git stash on the builder's side), bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_a; mkdir -p $O
( time timeout 1700 python -m pytest tests -m gpu -q ) > $O/tests.log 2>&1 < /dev/null; tail -4 $O/tests.log
timeout 300 python tools/gram_i8_bench.py > $O/gram.txt 2>&1; tail -2 $O/gram.txt
timeout 600 python bench.py --steps 8 --warmup 2 --cpu-sample 0 > $O/bench.json 2> $O/bench.err < /dev/null; cut -c1-300 $O/bench.json
