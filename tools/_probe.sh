cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fp8.py -x -q -s 2>&1 | tail -8
timeout 300 python bench.py --no-cpu-baseline --no-secondary --e2e-tiles 0 --fp8 --steps 4 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fp8 headline', d['value'], d['roofline']['time_share'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace -d /tmp/k8 -o k8 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --e2e-tiles 0 --fp8 > /tmp/k8.log 2>&1 < /dev/null
DB=$(find /tmp/k8 -name "*.db" | head -1); [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $R/gpurun_out/r03_rocprofv3_vitonly_kernel_stats_fp8.txt
head -12 $R/gpurun_out/r03_rocprofv3_vitonly_kernel_stats_fp8.txt | cut -c1-170
