# Round 6: the GPU test suite, smoke, then the round's evidence (tools/prof_round.sh r06) in one lease.
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06_gputests.log
tail -4 gpurun_out/r06_gputests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/prof_round.sh r06 2>&1 | tail -5
