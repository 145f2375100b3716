set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 400 python $R/bench.py --model ctranspath 2>/dev/null | tail -1 > $R/gpurun_out/bench_ctranspath.json
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/train_only.py > /tmp/kt.log 2>&1 < /dev/null
DB=$(find /tmp/kt -name "*.db" | head -1); [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/mil_train_stats.txt
timeout 300 rocprofv3 --kernel-trace -d /tmp/ks2 -o ks2 -- python $R/tools/swin_only.py 1024 1024 2 > /tmp/ks2.log 2>&1 < /dev/null
DB=$(find /tmp/ks2 -name "*.db" | head -1); [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/ctranspath_stats.txt
cut -c1-300 $R/gpurun_out/bench_ctranspath.json; head -5 $R/gpurun_out/mil_train_stats.txt | cut -c1-160; head -5 $R/gpurun_out/ctranspath_stats.txt | cut -c1-160
