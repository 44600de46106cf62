#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_nn; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -x -k pruned > $O/tests_pruned.log 2>&1 < /dev/null; tail -3 $O/tests_pruned.log
timeout 900 python -m pytest tests -m gpu -q -x -k "nn or kmeans or rowmin or half_precision or c3_subsample or landmarks or labels" > $O/tests_nn.log 2>&1 < /dev/null; tail -3 $O/tests_nn.log
bash tools/r06_nn_trace.sh | head -6
bash tools/r06_km_trace.sh | head -12
timeout 600 python tools/dropin_profile.py > $O/dropin.txt 2>&1; head -3 $O/dropin.txt; grep -n "kmeans\|nn_distances\|map_solve" $O/dropin.txt | head -4
