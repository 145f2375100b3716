# SQ counters of the T = 257 attention kernel (two passes: the SQ block takes ~8 counters at a time)
set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d /tmp/sq1 -o sq -- python $R/tools/attn_only.py 1020 257 16 > /tmp/sq1.log 2>&1 < /dev/null
tail -c 200 /tmp/sq1.log
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d /tmp/sq2 -o sq -- python $R/tools/attn_only.py 1020 257 16 > /tmp/sq2.log 2>&1 < /dev/null
tail -c 200 /tmp/sq2.log
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_WAVE32_LDS -d /tmp/sq3 -o sq -- python $R/tools/attn_only.py 1020 257 16 > /tmp/sq3.log 2>&1 < /dev/null
tail -c 200 /tmp/sq3.log
for i in 1 2 3; do
  DB=$(find /tmp/sq$i -name "*.db" | head -1); echo "db=$DB"
  [ -n "$DB" ] && timeout 60 python $R/tools/pmc_dump.py "$DB" attn
done > $R/gpurun_out/pmc_attn_sq.txt 2>&1
cat $R/gpurun_out/pmc_attn_sq.txt
