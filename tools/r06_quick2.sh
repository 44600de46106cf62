#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_quick; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_estimators.py tests/test_gpu_sharded.py tests/test_gpu_round5.py -m gpu -q -x > $O/tests.log 2>&1 < /dev/null; tail -2 $O/tests.log
bash tools/r06_quick.sh
