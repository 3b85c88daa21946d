TAG=r2sq
REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out; cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --warmup 3 --no-cpu-baseline --no-roofline --no-extras"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/${TAG}_pmc/SQ -o p -- $BENCH --steps 2 > $OUT/${TAG}_pmc.SQ.log 2>&1
tail -1 $OUT/${TAG}_pmc.SQ.log | cut -c1-100
cd $REPO && python tools/profile_summary.py $OUT $TAG > /dev/null; rm -rf $OUT/${TAG}_pmc
