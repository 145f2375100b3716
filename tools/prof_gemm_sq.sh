set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d /tmp/sq -o sq -- python $R/tools/gemm_model_shapes.py > /tmp/sq.log 2>&1 < /dev/null
tail -c 400 /tmp/sq.log
DB=$(find /tmp/sq -name "*.db" | head -1); echo "db=$DB"
[ -n "$DB" ] && timeout 60 python $R/tools/pmc_dump.py "$DB" gemm > $R/gpurun_out/pmc_gemm_sq.txt
cat $R/gpurun_out/pmc_gemm_sq.txt | head -60
