cd $GRAFT_REPO_ROOT
timeout 240 python -m pytest tests/test_gpu_tiling.py -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for w in 8 32; do timeout 120 python $GRAFT_REPO_ROOT/tools/slide_only.py 20736 vit_large_patch14_224 canny $w 64 2>&1 | tail -1 | cut -c1-640; done
timeout 120 python $GRAFT_REPO_ROOT/tools/slide_only.py 20736 vit_large_patch14_224 nocanny 32 64 2>&1 | tail -1 | cut -c1-640
