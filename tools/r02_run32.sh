set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd $R
run() { timeout 300 python bench.py --no-secondary --no-cpu-baseline --e2e-tiles 0 --tiles 2040 --steps 4 "$@" 2>&1 | grep -o '"value": [0-9.]*' | head -1; }
echo "base chunk 1020 no overlap"; run --chunk 1020 --overlap 0
for c in 96 128 160 192 224; do echo "overlap chunk 1020 attn cus $c"; AMDS_ATTN_CUS=$c run --chunk 1020 --overlap 1; done
echo "base chunk 1020 no overlap"; run --chunk 1020 --overlap 0
