# Full GPU check of the tree: parity tests, smoke, the default bench line.  Usage (from the container):
#   gpurun --timeout 2400 -- 'bash tools/gpu_full.sh'
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gputests.log
tail -5 gpurun_out/gputests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py 2> gpurun_out/bench_err.log | tail -1 > gpurun_out/bench_default.json
cut -c1-600 gpurun_out/bench_default.json
