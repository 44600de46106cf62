#!/bin/bash
# SQ counters of k_gram_i8 alone (tools/gram_i8_bench.py: 60 000 x 5000), three passes of <= 4 counters
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06_gram_pmc; mkdir -p $O
i=0
for set in "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd $GRAFT_REPO_ROOT && timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p$i -- python tools/gram_i8_bench.py > $O/p$i.log 2>&1 < /dev/null)
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find $O -name "*.db") --match k_gram_i8 > $O/r06_gram_i8_pmc.txt 2> $O/summary.err
find $O -name "*.db" -delete
cat $O/r06_gram_i8_pmc.txt
