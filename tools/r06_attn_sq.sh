#!/bin/bash
# SQ counters of the MIL heads' attention kernels inside the training step (two PMC passes, kernel-trace only beside them)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA -d /tmp/sq1 -o sq -- python $R/tools/train_only.py 4 > /tmp/sq1.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU_TRANS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/sq2 -o sq -- python $R/tools/train_only.py 4 > /tmp/sq2.log 2>&1 < /dev/null
for i in 1 2; do
  DB=$(find /tmp/sq$i -name "*.db" | head -1); echo "== pass $i"
  [ -n "$DB" ] && timeout 60 python $R/tools/pmc_dump.py "$DB" attn_flash attn_bwd
done > $R/gpurun_out/r06_pmc_mil_attn_sq.txt 2>&1
cat $R/gpurun_out/r06_pmc_mil_attn_sq.txt
