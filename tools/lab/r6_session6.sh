export TMPDIR=/tmp
O=gpurun_out/r6g; mkdir -p $O
timeout 1500 python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "every_non_default_variant or one_point_neighbourhoods or layer_tile_subset or weighted_statistics or dedup" > $O/pair_tests.txt 2>&1; tail -8 $O/pair_tests.txt
AB_STEPS=60 bash tools/lab/ab_opts.sh "-" "PAIRED_LAUNCHES=0" > $O/ab_pair.txt 2>&1; cat $O/ab_pair.txt
AB_STEPS=40 AB_ARGS="--neighbourhoods whole" bash tools/lab/ab_opts.sh "-" "PAIRED_LAUNCHES=0" > $O/ab_pair_whole.txt 2>&1; cat $O/ab_pair_whole.txt
