#!/bin/bash
# SQ counters of the fp16-split row-minimum sweep (1-NN pre-filter: the one launch above 100 ms of tools/nn_probe_c3.py)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/rowmin_pmc; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CYCLES SQ_CYCLES"; do
  i=$((i+1))
  (cd $GRAFT_REPO_ROOT && timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p$i -- python tools/nn_probe_c3.py > $O/p$i.log 2>&1 < /dev/null)
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find $O -name "*.db") --match k_rowmin_f16x3 --min-us 100000 > $O/r05_rowmin_pmc.txt 2> $O/summary.err
find $O -name "*.db" -delete
head -40 $O/r05_rowmin_pmc.txt; tail -2 $O/p1.log
