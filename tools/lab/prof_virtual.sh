cd /tmp && export TMPDIR=/tmp
PDR_VIRTUAL_FIRST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/profv -o rv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/profv.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/profv.log | cut -c1-160
