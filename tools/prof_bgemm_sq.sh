# SQ counters of the fp32 batched GEMM on two shapes (big wgrad-like product; 256^3 pinv product)
set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for shape in "64 1536 512 1280" "512 256 256 256"; do
  for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $pmc -d /tmp/bq$i -o bq -- python $R/tools/bgemm_f32_one.py $shape > /tmp/bq$i.log 2>&1 < /dev/null
    DB=$(find /tmp/bq$i -name "*.db" | head -1)
    echo "== shape $shape =="
    [ -n "$DB" ] && timeout 60 python $R/tools/pmc_dump.py "$DB" bgemm || tail -5 /tmp/bq$i.log
  done
done > $R/gpurun_out/pmc_bgemm_sq.txt 2>&1
cat $R/gpurun_out/pmc_bgemm_sq.txt
