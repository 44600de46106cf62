#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c; mkdir -p $O
timeout 300 python tools/gram_i8_bench.py > $O/gram.txt 2>&1; tail -2 $O/gram.txt
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -s > $O/tests6.log 2>&1 < /dev/null; tail -12 $O/tests6.log
timeout 900 python -m pytest tests -m gpu -q -x -k "gram or precond or golden or rebuild or sharded" > $O/tests_gram.log 2>&1 < /dev/null; tail -3 $O/tests_gram.log
bash tools/r06_gram_pmc.sh > $O/gram_pmc.log 2>&1; tail -14 $O/gram_pmc.log
