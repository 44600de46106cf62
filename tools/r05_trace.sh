#!/bin/bash
# kernel trace of three C3 steps (tools/one_step.py) -> timeline + per-kernel totals of the last step
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/trace_$1; mkdir -p $O
rocprofv3 --kernel-trace -d $O/db -o one -- python tools/one_step.py > $O/one.log 2>&1
DB=$(find $O/db -name "*.db" | head -1)
python tools/step_timeline.py $DB > $O/step_timeline.txt 2> $O/tl.err
python tools/step_kernels.py $DB > $O/step_kernels.txt 2> $O/sk.err
find $O -name "*.db" -delete
head -40 $O/step_kernels.txt
