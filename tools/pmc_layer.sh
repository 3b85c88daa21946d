# usage: bash tools/pmc_layer.sh <shape index> <tag>   (one SQ counter pass on the layer micro-benchmark)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT -o p -- python -m tools.fused_layer_bench --only $1 --reps 5 > $OUT.log 2>&1
tail -3 $OUT.log
ls $OUT
