#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_km; mkdir -p $O
KM_D=20 bash tools/r06_km_trace.sh | head -12
MELLON_AMD_EXPERIMENTAL=1 timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -x -k "group_bounds" > $O/tests_gb.log 2>&1 < /dev/null; tail -3 $O/tests_gb.log
timeout 600 python tools/dropin_profile.py > $O/dropin.txt 2>&1; head -3 $O/dropin.txt; grep -n "kmeans\|nn_distances\|map_solve" $O/dropin.txt | head -4
