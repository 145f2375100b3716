# round-2 GPU call 1: widened tile-encoder parity + ViT-only kernel trace
set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
nproc > gpurun_out/r02_nproc.txt; lscpu | head -20 >> gpurun_out/r02_nproc.txt
timeout 1500 python -m pytest tests/test_gpu_vit.py -x -q -m gpu -s 2>&1 | tail -40 > gpurun_out/r02_pytest_vit.log
cat gpurun_out/r02_pytest_vit.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d /tmp/ks -o ks -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/ks.log 2>&1 < /dev/null
tail -c 3000 /tmp/ks.log > $R/gpurun_out/r02_vitonly_bench.json
DB=$(find /tmp/ks -name "*.db" | head -1); echo "db=$DB"
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r02_vitonly_kernel_stats.txt
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" --by-shape > $R/gpurun_out/r02_vitonly_kernel_stats_by_shape.txt
head -30 $R/gpurun_out/r02_vitonly_kernel_stats_by_shape.txt
