# round-2 GPU call 3: full GPU suite + default bench line
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( time timeout 3000 python -m pytest tests -q -m gpu -s 2>&1 | tail -250 ) > gpurun_out/r02_pytest_all.log 2>&1
grep -E "passed|failed" gpurun_out/r02_pytest_all.log | tail -5
( time timeout 900 python bench.py ) > gpurun_out/r02_bench_default.log 2>&1
tail -c 6000 gpurun_out/r02_bench_default.log
