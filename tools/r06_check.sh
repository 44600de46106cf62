#!/bin/bash
# health check of a build: the whole GPU suite, smoke, a short bench line (no CPU baseline)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_check; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/tests.log 2>&1 < /dev/null; tail -6 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 < /dev/null; tail -1 $O/smoke.log
timeout 600 python bench.py --steps 8 --warmup 2 --cpu-sample 0 > $O/bench.json 2> $O/bench.err < /dev/null; cut -c1-250 $O/bench.json
