#!/bin/bash
# SQ counters of the one-wave-per-SIMD row-minimum sweep (1-NN pre-filter: the launch above 100 ms of tools/nn_probe_c3.py)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06_rowmin_pmc; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  (cd $GRAFT_REPO_ROOT && timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p$i -- python tools/nn_probe_c3.py > $O/p$i.log 2>&1 < /dev/null)
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find $O -name "*.db") --match k_rowmin_w64 --min-us 100000 > $O/r06_rowmin_pmc.txt 2> $O/summary.err
find $O -name "*.db" -delete
cat $O/r06_rowmin_pmc.txt; tail -2 $O/p1.log
