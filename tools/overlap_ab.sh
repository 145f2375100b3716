# A/B of the two-chunk overlapped schedule (amds_vit_forward_overlapped) on 2040 tiles:  gpurun -- bash tools/overlap_ab.sh   (measured: 5 643 vs 5 717 tiles/s)
cd $GRAFT_REPO_ROOT
for ov in 0 1; do timeout 300 python bench.py --no-cpu-baseline --no-secondary --e2e-tiles 0 --tiles 2040 --chunk 1020 --overlap $ov --steps 4 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('overlap $ov', d['value'])"; done
