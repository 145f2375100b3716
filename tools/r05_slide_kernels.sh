cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace -d /tmp/sl -o sl -- python $R/tools/slide_only.py 12288 vit_large_patch14_224 canny 32 64 > /tmp/sl.log 2>&1 < /dev/null
DB=$(find /tmp/sl -name "*.db" | head -1); python $R/tools/rocprof_summary.py "$DB" | head -16 | cut -c1-60,108-175
tail -2 /tmp/sl.log
