#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_e; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "nn or kmeans or rowmin or half_precision or c3_subsample or landmarks or labels" > $O/tests_nn.log 2>&1 < /dev/null; tail -5 $O/tests_nn.log
timeout 300 python tools/nn_probe_c3.py > $O/nn_new.txt 2>&1; tail -2 $O/nn_new.txt
MELLON_AMD_EXPERIMENTAL=1 MELLON_AMD_ROWMIN_W64=0 timeout 300 python tools/nn_probe_c3.py > $O/nn_old.txt 2>&1; tail -2 $O/nn_old.txt
timeout 600 python tools/dropin_profile.py > $O/dropin.txt 2>&1; head -30 $O/dropin.txt
