cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 300 rocprofv3 --kernel-trace -d /tmp/gp -o gp -- python $R/tools/scratch/gap_prof.py > /tmp/gp.log 2>&1 < /dev/null
tail -c 400 /tmp/gp.log
DB=$(find /tmp/gp -name "*.db" | head -1); echo "db=$DB"
python $R/tools/rocprof_summary.py "$DB" --by-shape 2>/dev/null | grep -E "kernel|gap|gemm_f32" | head -40 || python $R/tools/rocprof_summary.py "$DB" | head -30
