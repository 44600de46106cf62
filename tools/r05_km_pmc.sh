#!/bin/bash
# SQ counters of the kernel-matrix pass alone (tools/km_probe.py), several passes of <= 4 counters each.
# usage: r05_km_pmc.sh [first-set last-set]   (default: all seven); only the n x m launches (>= 5 ms) are averaged.
# GRBM_GUI_ACTIVE / 8 XCDs / duration = the shader clock during the pass.
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/km_pmc; mkdir -p $O
A=${1:-1}; B=${2:-7}
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CYCLES SQ_CYCLES"; do
  i=$((i+1))
  [ $i -lt $A ] && continue; [ $i -gt $B ] && continue
  (cd $GRAFT_REPO_ROOT && timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p$i -- python tools/km_probe.py > $O/p$i.log 2>&1 < /dev/null)
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find $O -name "*.db") --match k_kernel_matrix_rows --min-us 5000 > $O/r05_km_pmc.txt 2> $O/summary.err
find $O -name "*.db" -delete
head -60 $O/r05_km_pmc.txt
