#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_i; mkdir -p $O
timeout 300 python tools/nn_probe_c3.py > $O/nn_w32.txt 2>&1; tail -2 $O/nn_w32.txt
MELLON_AMD_EXPERIMENTAL=1 MELLON_AMD_ROWMIN_SHAPE=64 timeout 300 python tools/nn_probe_c3.py > $O/nn_w64.txt 2>&1; tail -2 $O/nn_w64.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "nn or kmeans or rowmin or half_precision or c3_subsample or landmarks or labels" > $O/tests_nn.log 2>&1 < /dev/null; tail -3 $O/tests_nn.log
timeout 600 python tools/dropin_profile.py > $O/dropin.txt 2>&1; head -22 $O/dropin.txt | tail -14
sed -i 's/--match k_rowmin_w64/--match k_rowmin_w/' tools/r06_rowmin_pmc.sh
bash tools/r06_rowmin_pmc.sh > $O/pmc.log 2>&1; tail -24 $O/pmc.log
