cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_mil_seam.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 120 python $R/tools/train_only.py 8 2>&1 | tail -1
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/train_only.py 8 > /tmp/kt.log 2>&1 < /dev/null
DB=$(find /tmp/kt -name "*.db" | head -1); [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/mil_train_stats.txt
head -12 $R/gpurun_out/mil_train_stats.txt | cut -c1-170
