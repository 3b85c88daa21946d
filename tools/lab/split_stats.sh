mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/split_stats -o s -- python $R/bench.py --warmup 3 --steps 20 --no-cpu-baseline --no-roofline --no-extras --precision split_f16 > $R/gpurun_out/split_stats.log 2>&1
tail -1 $R/gpurun_out/split_stats.log | cut -c1-200
cp $(find $R/gpurun_out/split_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/split_kernel_stats.csv; rm -rf $R/gpurun_out/split_stats
