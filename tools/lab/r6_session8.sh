export TMPDIR=/tmp
O=gpurun_out/r6i; mkdir -p $O
AB_STEPS=60 bash tools/lab/ab_opts.sh "-" "AHEAD_ENCODER_MAPS=0" > $O/ab_enc_hoist.txt 2>&1; cat $O/ab_enc_hoist.txt
timeout 1500 python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "every_non_default_variant or one_point_neighbourhoods or adaptive_sampler" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
