# round-2 GPU call 61: tiles per forward chunk, alternating (is 4080 better than 1020 beyond drift?)
set -x
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2 3; do
  for c in 1020 4080 2040; do
    echo "chunk=$c"
    timeout 400 python bench.py --tiles $c --chunk $c --steps 4 --warmup 1 --no-secondary --no-cpu-baseline --e2e-tiles 0 2>&1 | grep -o '"value": [0-9.]*, "unit": "tiles/s", "n_gpus"'
  done
done
