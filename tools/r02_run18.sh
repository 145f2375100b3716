set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
for sp in 0 100 200 400 0; do
AMDS_GEMM_STAGGER_PROJ=$sp AMDS_GEMM_STAGGER_FC2=$((sp*2)) timeout 300 python bench.py --no-secondary --no-cpu-baseline --e2e-tiles 0 2>&1 | grep -o '"value": [0-9.]*' | head -1
done > gpurun_out/r02_run18.log 2>&1
cat gpurun_out/r02_run18.log
