#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -s -p no:xdist > $O/tests6.log 2>&1 < /dev/null; tail -15 $O/tests6.log
bash tools/r06_gram_pmc.sh > $O/gram_pmc.log 2>&1; tail -16 $O/gram_pmc.log
bash tools/r06_cfg_trace.sh c2 > $O/c2.log 2>&1; tail -50 $O/c2.log
