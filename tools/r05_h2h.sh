#!/bin/bash
# where the host-to-host step's extra milliseconds go: cProfile + stage times of one C3 fit_predict from device / page-locked /
# pageable cells (tools/cprofile_fit.py), and the plain copy rates (tools/h2d_probe.py)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_h2h; mkdir -p $O
for mode in device pinned host; do
  timeout 300 python tools/cprofile_fit.py $mode > $O/$mode.txt 2> $O/$mode.err < /dev/null
done
timeout 200 python tools/h2d_probe.py > $O/h2d.txt 2>&1 < /dev/null
grep "step ms" $O/*.txt; grep -h kernel_matrix_s $O/*.txt | cut -c1-60
