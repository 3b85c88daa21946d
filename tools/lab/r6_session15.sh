export TMPDIR=/tmp
O=gpurun_out/r6p; mkdir -p $O
timeout 1500 python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "every_non_default_variant or adaptive_sampler" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
AB_STEPS=60 bash tools/lab/ab_opts.sh "-" "QUERY_CONV_AHEAD=0" > $O/ab.txt 2>&1; cat $O/ab.txt
AB_STEPS=40 AB_ARGS="--neighbourhoods whole" bash tools/lab/ab_opts.sh "-" "QUERY_CONV_AHEAD=0" > $O/ab_whole.txt 2>&1; cat $O/ab_whole.txt
