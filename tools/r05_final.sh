#!/bin/bash
# round-5 closing run: the whole GPU suite, smoke, the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 < /dev/null; tail -2 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 < /dev/null; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null; cut -c1-400 $O/bench.json
