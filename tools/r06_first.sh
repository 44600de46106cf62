#!/bin/bash
# round-6 opening run: GPU suite, default bench line (short CPU baseline), host-side timeline of one step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_first; mkdir -p $O
( time timeout 1700 python -m pytest tests -m gpu -x -q ) > $O/tests.log 2>&1 < /dev/null; tail -4 $O/tests.log
timeout 600 python bench.py --steps 8 --warmup 2 --cpu-sample 12000 > $O/bench.json 2> $O/bench.err < /dev/null; cut -c1-600 $O/bench.json
MELLON_AMD_TRACE=1 timeout 300 python tools/one_step.py > $O/trace.log 2>&1 < /dev/null; tail -60 $O/trace.log
