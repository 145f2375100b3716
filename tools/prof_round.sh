# Evidence of a round, one gpurun call:  gpurun --timeout 3000 -- 'bash tools/prof_round.sh r04'
# Writes gpurun_out/<tag>_*; copy what should be judged into profiles/.
TAG=${1:-r06}
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the default bench line (all legs)
timeout 900 python $R/bench.py 2> $O/${TAG}_bench_err.log | tail -1 > $O/${TAG}_bench_default.json
# 2. ViT-only kernel trace: totals and per GEMM shape
timeout 400 rocprofv3 --kernel-trace -d /tmp/ks -o ks -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --e2e-tiles 0 > /tmp/ks.log 2>&1 < /dev/null
DB=$(find /tmp/ks -name "*.db" | head -1)
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $O/${TAG}_rocprofv3_vitonly_kernel_stats.txt
[ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" --by-shape 2 > $O/${TAG}_rocprofv3_vitonly_kernel_stats_by_shape.txt
# 3. HBM traffic of the GEMM: separate PMC passes
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --e2e-tiles 0 > /tmp/pf.log 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pw -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --e2e-tiles 0 > /tmp/pw.log 2>&1 < /dev/null
F=$(find /tmp/pf -name "*.db" | head -1); W=$(find /tmp/pw -name "*.db" | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then
  timeout 60 python $R/tools/pmc_summary.py "$F" "$W" gemm attn ln_ im2col > $O/${TAG}_pmc_hbm_traffic.txt
  ALG=$(python -c "import json; print(json.load(open('$O/${TAG}_bench_default.json'))['roofline']['algorithmic_bytes_per_launch'])")
  (cd $R/tools && timeout 60 python pmc_traffic_json.py "$F" "$W" $ALG > $O/${TAG}_pmc_gemm_traffic.json)
fi
# 4. the opt-in modes and the other presets
timeout 400 rocprofv3 --kernel-trace -d /tmp/kx -o kx -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --e2e-tiles 0 --exact > /tmp/kx.log 2>&1 < /dev/null
DB=$(find /tmp/kx -name "*.db" | head -1); [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $O/${TAG}_rocprofv3_vitonly_kernel_stats_exact.txt
timeout 400 rocprofv3 --kernel-trace -d /tmp/k8 -o k8 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --e2e-tiles 0 --fp8 > /tmp/k8.log 2>&1 < /dev/null
DB=$(find /tmp/k8 -name "*.db" | head -1); [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $O/${TAG}_rocprofv3_vitonly_kernel_stats_fp8.txt
timeout 300 python $R/tools/gemm_fp8_bench.py > $O/${TAG}_gemm_fp8_bench.txt 2>&1
for m in uni2_h virchow2; do
  timeout 500 python $R/bench.py --model $m --no-secondary --e2e-tiles 0 --cpu-seconds 8 2>> $O/${TAG}_bench_err.log | tail -1 > $O/${TAG}_bench_$m.json
done
# 5. MIL training step, slide pipeline
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/train_only.py 8 > /tmp/kt.log 2>&1 < /dev/null
DB=$(find /tmp/kt -name "*.db" | head -1); [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" > $O/${TAG}_rocprofv3_mil_train_kernel_stats.txt
tail -1 /tmp/kt.log >> $O/${TAG}_rocprofv3_mil_train_kernel_stats.txt
timeout 300 rocprofv3 --kernel-trace -d /tmp/sl -o sl -- python $R/tools/slide_only.py 12288 vit_large_patch14_224 canny 32 64 > /tmp/sl.log 2>&1 < /dev/null
DB=$(find /tmp/sl -name "*.db" | head -1); [ -n "$DB" ] && (cd $R/tools && timeout 60 python rocprof_gaps.py "$DB" 1000 > $O/${TAG}_slide_pipeline_gaps.txt); tail -1 /tmp/sl.log >> $O/${TAG}_slide_pipeline_gaps.txt
# 6. gated-attention pooling: A/B of the fused launch against the six-launch form, kernel trace of both
timeout 300 python $R/tools/gap_only.py > $O/${TAG}_gap_ab.json 2>> $O/${TAG}_bench_err.log
timeout 300 rocprofv3 --kernel-trace -d /tmp/kg -o kg -- python $R/tools/gap_only.py --trace > /tmp/kg.log 2>&1 < /dev/null
DB=$(find /tmp/kg -name "*.db" | head -1); [ -n "$DB" ] && timeout 60 python $R/tools/rocprof_summary.py "$DB" --by-shape > $O/${TAG}_gap_kernel_trace.txt
# 7. TransMIL training step
bash $R/tools/r06_transmil_trace.sh
cut -c1-400 $O/${TAG}_bench_default.json
