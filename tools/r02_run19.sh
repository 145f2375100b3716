set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( time timeout 1200 python bench.py ) > gpurun_out/r02_bench_default.log 2>&1
tail -c 2500 gpurun_out/r02_bench_default.log
