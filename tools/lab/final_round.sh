set -x
cd $GRAFT_REPO_ROOT
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r6_final_tests.log 2>&1
tail -3 gpurun_out/r6_final_tests.log
bash tools/profile_round.sh r6 > gpurun_out/r6_profile_round.log 2>&1
tail -5 gpurun_out/r6_profile_round.log
( time python bench.py ) > gpurun_out/r6_bench_full.log 2> gpurun_out/r6_bench_full.err
tail -c 600 gpurun_out/r6_bench_full.log
MARK_BACK_TO_BACK=6 python -m tools.lab.step_markers gpurun_out/r6_timeline_markers_loop.json > gpurun_out/r6_timeline_markers_loop.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
