set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
for sp in 0 100 200 400 800; do
echo "stagger proj=$sp fc2=$((sp*2))"
AMDS_GEMM_STAGGER_PROJ=$sp AMDS_GEMM_STAGGER_FC2=$((sp*2)) timeout 300 python tools/lnfold_producer_ab.py 2>&1 | grep "producer form" | sed -n '2p;4p'
done > gpurun_out/r02_run26.log 2>&1
cat gpurun_out/r02_run26.log
