# round-2 GPU call 57: chunk size sweep of the tile encoder (tiles per forward chunk)
set -x
R=$GRAFT_REPO_ROOT; cd $R
for c in 1020 2040 4080 510 1020; do
  echo "chunk=$c"
  timeout 400 python bench.py --tiles $c --chunk $c --steps 4 --warmup 1 --no-secondary --no-cpu-baseline --e2e-tiles 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "tiles/s", "n_gpus"'
done
