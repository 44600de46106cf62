#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_g; mkdir -p $O
timeout 300 python tools/nn_probe_c3.py > $O/nn_new.txt 2>&1; tail -2 $O/nn_new.txt
timeout 300 python tools/gram_i8_bench.py > $O/gram.txt 2>&1; tail -1 $O/gram.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "nn or kmeans or rowmin or half_precision or c3_subsample or landmarks or labels or gram or precond" > $O/tests_nn.log 2>&1 < /dev/null; tail -3 $O/tests_nn.log
bash tools/r06_rowmin_pmc.sh > $O/pmc.log 2>&1; tail -24 $O/pmc.log
timeout 600 python tools/dropin_profile.py > $O/dropin.txt 2>&1; head -22 $O/dropin.txt
