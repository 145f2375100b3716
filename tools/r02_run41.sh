# round-2 GPU call 41: is TransMIL training still at ~750 bags/s?  (345 in the last full run) -- old / new library, kernel trace
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
nproc; uptime
for L in build/ab/libamdstamp_old.so stamp_amd/lib/libamdstamp.so build/ab/libamdstamp_old.so stamp_amd/lib/libamdstamp.so; do
  echo "lib=$L"
  AMDSTAMP_LIB=$R/$L timeout 200 python tools/transmil_train_only.py 64 6 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/transmil_train_only.py 64 3 > /tmp/kt.log 2>&1 < /dev/null
tail -2 /tmp/kt.log
DB=$(find /tmp/kt -name "*.db" | head -1)
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" | head -25
