cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fp8.py -x -q -s -k vit_fp8 2>&1 | tail -12
timeout 300 python bench.py --no-cpu-baseline --no-secondary --e2e-tiles 0 --fp8 --steps 4 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fp8 headline', d['value'], d['roofline']['time_share'])"
