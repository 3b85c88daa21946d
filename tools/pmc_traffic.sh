# HBM traffic of the bench's kernels from the L2 memory-side counters: two separate --pmc passes
# (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2), kernel-trace only (MI355X_MICROARCH.md, HBM section).
# usage (GPU box):  bash tools/pmc_traffic.sh <tag>     -> gpurun_out/pmc_traffic_<tag>/{fetch,write}/...
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic_$1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT.$c.log 2>&1
  tail -1 $OUT.$c.log | cut -c1-120
done
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py $OUT
