set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d /tmp/kb -o kb -- python $R/tools/b64_only.py 4 > /tmp/kb.log 2>&1 < /dev/null
tail -1 /tmp/kb.log
DB=$(find /tmp/kb -name "*.db" | head -1); echo "db=$DB"
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" --seq 430 > $R/gpurun_out/r02_b64_sequence.txt
tail -3 $R/gpurun_out/r02_b64_sequence.txt
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" | head -14 | cut -c1-80,110-200
