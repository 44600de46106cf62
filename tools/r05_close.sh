#!/bin/bash
# closing measurements of round 5 at one commit: rocprofv3 summary + PMC traffic of the default bench command, then the bench
# lines / emulated ranks / timeline / drop-in profile (tools/r05_batch.sh), the kernel-matrix counters and the copy rates
cd $GRAFT_REPO_ROOT
COMMIT=$1 bash tools/profile_round.sh r05 > gpurun_out/prof_r05.log 2>&1 < /dev/null
bash tools/r05_batch.sh > gpurun_out/r05_batch.log 2>&1 < /dev/null
bash tools/r05_h2h.sh > gpurun_out/r05_h2h.log 2>&1 < /dev/null
tail -12 gpurun_out/r05_batch.log; tail -5 gpurun_out/prof_r05.log
