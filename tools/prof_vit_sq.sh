# SQ counters of the tile encoder's kernels in the bench run itself (ViT-L/14, one 1020-tile chunk, LayerNorm folded): MFMA pipe busy, VALU / LDS activity, waits
set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $pmc -d /tmp/vq$i -o vq -- python $R/bench.py --steps 1 --warmup 1 --tiles 1020 --no-cpu-baseline --no-secondary --e2e-tiles 0 > /tmp/vq$i.log 2>&1 < /dev/null
  DB=$(find /tmp/vq$i -name "*.db" | head -1)
  [ -n "$DB" ] && timeout 60 python $R/tools/pmc_dump.py "$DB" gemm_4w16 attn_vit257 ln_stats || tail -5 /tmp/vq$i.log
done > $R/gpurun_out/r02_pmc_vit_sq.txt 2>&1
cat $R/gpurun_out/r02_pmc_vit_sq.txt
