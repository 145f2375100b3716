#!/bin/bash
# NOTE: A/B against a second library stamp_amd/lib/libamdstamp_<tag>.so built from the previous commit (AMDSTAMP_LIB selects it); profiles/r06_attn_bwd_ab.txt.
# A/B of attention-backward builds: default library vs stamp_amd/lib/libamdstamp_<tag>.so (AMDSTAMP_LIB), alternating; then a kernel trace of each with the by-shape table
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-dqB}
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/r06_attn_ab_${TAG}.txt
: > $OUT
( cd $R && AMDSTAMP_LIB=$R/stamp_amd/lib/libamdstamp_${TAG}.so timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_mil_seam.py -x -q 2>&1 | tail -2 | sed "s/^/$TAG tests: /" ) >> $OUT
for rep in 1 2 3; do
  ( cd $R && timeout 200 python tools/train_only.py 30 | tail -1 | sed "s/^/default /"; AMDSTAMP_LIB=$R/stamp_amd/lib/libamdstamp_${TAG}.so timeout 200 python tools/train_only.py 30 | tail -1 | sed "s/^/$TAG /" ) >> $OUT
done
for v in default $TAG; do
  L=$R/stamp_amd/lib/libamdstamp.so; [ $v != default ] && L=$R/stamp_amd/lib/libamdstamp_${v}.so
  rm -rf /tmp/kt; AMDSTAMP_LIB=$L timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/train_only.py 8 > /tmp/kt.log 2>&1 < /dev/null
  DB=$(find /tmp/kt -name "*.db" | head -1)
  [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r06_attn_ab_${v}_kernel_stats.txt
  [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" --by-shape > $R/gpurun_out/r06_attn_ab_${v}_by_shape.txt
  grep -E "attn_|TOTAL" $R/gpurun_out/r06_attn_ab_${v}_kernel_stats.txt | cut -c1-60,110-180 | sed "s/^/$v /" >> $OUT
done
cat $OUT
