#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_km; mkdir -p $O
KM_D=20 bash tools/r06_km_trace.sh | head -16
timeout 1200 python -m pytest tests -m gpu -q -x -k "kmeans or landmarks or labels or one_upload or group_bounds" > $O/tests_km.log 2>&1 < /dev/null; tail -5 $O/tests_km.log
timeout 600 python tools/dropin_profile.py > $O/dropin.txt 2>&1; head -3 $O/dropin.txt; grep -n "kmeans\|nn_distances\|map_solve" $O/dropin.txt | head -4
