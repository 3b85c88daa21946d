cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof19 -o r19 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof19.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof19.log | cut -c1-160
