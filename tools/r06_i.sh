#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_i; mkdir -p $O
timeout 300 python tools/nn_probe_c3.py > $O/nn_new.txt 2>&1; tail -2 $O/nn_new.txt
timeout 300 python tools/km_cmp.py > $O/km_new.txt 2>&1; tail -3 $O/km_new.txt
MELLON_AMD_ROWMIN_W64=0 timeout 300 python tools/km_cmp.py > $O/km_old.txt 2>&1; tail -3 $O/km_old.txt
