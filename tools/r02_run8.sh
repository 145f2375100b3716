# round-2 GPU call 8: residual-epilogue prefetch A/B -- GEMM / ViT tests, ViT-only bench line, per-shape kernel trace
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vit.py tests/test_gpu_train.py -q -m gpu -x 2>&1 | tail -15 ) > gpurun_out/r02_run8_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/r02_run8_pytest.log | tail -3
timeout 600 python bench.py --no-secondary --no-cpu-baseline --e2e-tiles 0 > gpurun_out/r02_run8_bench.log 2>&1
tail -c 900 gpurun_out/r02_run8_bench.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d /tmp/ks -o ks -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --e2e-tiles 0 > /tmp/ks.log 2>&1 < /dev/null
DB=$(find /tmp/ks -name "*.db" | head -1); echo "db=$DB"
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" --by-shape > $R/gpurun_out/r02_run8_by_shape.txt
head -16 $R/gpurun_out/r02_run8_by_shape.txt | cut -c1-60,100-200
