#!/bin/bash
# per-kernel times of the headline step with the fused qkv + attention kernel on and off (rocprofv3 kernel trace, ViT only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for v in 1 0; do
  rm -rf /tmp/kq$v
  AMDS_VIT_QKVATTN=$v timeout 400 rocprofv3 --kernel-trace -d /tmp/kq$v -o kq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --e2e-tiles 0 --slide-tiles 0 > /tmp/kq$v.log 2>&1 < /dev/null
  DB=$(find /tmp/kq$v -name "*.db" | head -1)
  [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r05_qa_insitu_kernels_$v.txt
  echo "== AMDS_VIT_QKVATTN=$v"; head -14 $R/gpurun_out/r05_qa_insitu_kernels_$v.txt | cut -c1-70,111-175
done
