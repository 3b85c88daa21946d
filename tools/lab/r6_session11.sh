export TMPDIR=/tmp
O=gpurun_out/r6l; mkdir -p $O
timeout 1500 python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "every_non_default_variant or adaptive_sampler" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
AB_STEPS=60 bash tools/lab/ab_opts.sh "-" "SPLIT_SOURCE_TABLES=0" > $O/ab_tables.txt 2>&1; cat $O/ab_tables.txt
AB_STEPS=40 AB_ARGS="--neighbourhoods whole" bash tools/lab/ab_opts.sh "-" "SPLIT_SOURCE_TABLES=0" > $O/ab_tables_whole.txt 2>&1; cat $O/ab_tables_whole.txt
