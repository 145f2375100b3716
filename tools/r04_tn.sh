set -x
cd $GRAFT_REPO_ROOT
python tools/wgrad_tn_bench.py 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY -d /tmp/pm -o pm -- python $GRAFT_REPO_ROOT/tools/wgrad_tn_bench.py 65600 > /tmp/pm.log 2>&1 < /dev/null
DB=$(find /tmp/pm -name "*.db" | head -1)
[ -n "$DB" ] && timeout 60 python $GRAFT_REPO_ROOT/tools/pmc_dump.py "$DB" gemm_4w16 | head -40
