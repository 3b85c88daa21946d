#!/bin/bash
# One command -> every profile artefact of a round (run on the GPU box):
#     bash tools/profile_round.sh r2            # writes gpurun_out/r2_*; copy the summaries into profiles/
#   1. rocprofv3 --kernel-trace --stats of `bench.py --steps 20 --warmup 3` (headline leg only)
#   2. three SEPARATE --pmc passes of the same command at --steps 2 (kernel-trace only, as the guide prescribes):
#      FETCH_SIZE, WRITE_SIZE (HBM traffic; FETCH_SIZE x2 gfx950 correction in the summary) and the SQ pass with
#      SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES (MFMA utilisation of every kernel of the bench)
#   3. tools/op_roofline.py: FPS / ball_query / kNN / Chamfer / EMD at the BASELINE sizes, HIP-event timed
#   4. tools/profile_summary.py -> <tag>_bench_kernel_stats.csv, <tag>_pmc_traffic.json, <tag>_mfma_util.json,
#      <tag>_roofline.json
TAG=${1:-r4}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --warmup 3 --no-cpu-baseline --no-roofline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o s -- $BENCH --steps 20 > $OUT/${TAG}_stats.log 2>&1
tail -1 $OUT/${TAG}_stats.log | cut -c1-200
# the same with the two halves of every block on ONE stream: per-kernel durations of kernels that have the chip to
# themselves (the figure a kernel roofline is about; bench.py's `roofline` is measured the same way)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats_serial -o s -- $BENCH --single-stream --steps 20 > $OUT/${TAG}_stats_serial.log 2>&1
cp $(find $OUT/${TAG}_stats_serial -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats_serial.csv
rm -rf $OUT/${TAG}_stats_serial
if [ -n "$PDR_PROFILE_QUICK" ]; then   # kernel stats + step timeline only (no counter passes)
  cd $REPO && python tools/profile_summary.py $OUT $TAG && rm -rf $OUT/${TAG}_stats
  exit 0
fi
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc/$c -o p -- $BENCH --steps 2 > $OUT/${TAG}_pmc.$c.log 2>&1
  tail -1 $OUT/${TAG}_pmc.$c.log | cut -c1-120
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD \
  --output-format csv -d $OUT/${TAG}_pmc/SQ -o p -- $BENCH --steps 2 > $OUT/${TAG}_pmc.SQ.log 2>&1
tail -1 $OUT/${TAG}_pmc.SQ.log | cut -c1-120
cd $REPO
python tools/op_roofline.py > $OUT/${TAG}_op_roofline.json 2> $OUT/${TAG}_op_roofline.err
# time stamps inside an UNTRACED graph replay (pdr_mark_time kernels captured with the step): block-level and, in the
# second file, inside every block
python -m tools.lab.step_markers $OUT/${TAG}_timeline_markers.json > $OUT/${TAG}_timeline_markers.txt 2>&1
MARK_DETAIL=1 python -m tools.lab.step_markers $OUT/${TAG}_timeline_markers_detail.json > /dev/null 2>&1
# the same for the opt-in split-f16 step (is the FPS chain still hidden behind a shorter first block?)
MARK_PRECISION=split_f16 python -m tools.lab.step_markers $OUT/${TAG}_timeline_markers_split.json > $OUT/${TAG}_timeline_markers_split.txt 2>&1
# the once-per-batch first step by itself (no vendor kernels since round 4)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_first -o s -- python $REPO/tools/first_step_profile.py 32 > $OUT/${TAG}_first_step.log 2>&1 )
cp $(find $OUT/${TAG}_first -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_first_step_kernel_stats.csv; rm -rf $OUT/${TAG}_first
# clocks and power while the headline loop runs (what the 'implied clock' of the MFMA-utilisation table is about)
( python bench.py --steps 990 --warmup 5 --no-cpu-baseline --no-roofline --no-extras > /dev/null 2>&1 & BP=$!
  # (990 timed steps = 8.6 s; sampled from process start: the loaded samples are the ones that count)
  for i in $(seq 1 30); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power (W)" | tr '\n' ' '; echo; sleep 0.3; done > $OUT/${TAG}_clocks_power.txt; wait $BP )
python tools/profile_summary.py $OUT $TAG
# raw counter dumps are large: keep only the summaries
rm -rf $OUT/${TAG}_pmc $OUT/${TAG}_stats
