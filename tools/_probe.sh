cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in 8 16 32 64; do timeout 300 python $R/tools/slide_only.py 12288 vit_large_patch14_224 canny $w 16 2>&1 | tail -1 | cut -c1-260; done
timeout 300 python $R/tools/slide_only.py 12288 vit_large_patch14_224 canny 32 64 2>&1 | tail -1 | cut -c1-260
timeout 300 python $R/tools/slide_only.py 12288 vit_large_patch14_224 canny 16 4 2>&1 | tail -1 | cut -c1-260
