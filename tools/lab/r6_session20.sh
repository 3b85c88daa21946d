export TMPDIR=/tmp
O=gpurun_out/r6v; mkdir -p $O
AB_STEPS=40 AB_ARGS="--neighbourhoods whole" bash tools/lab/ab_opts.sh "-" "AHEAD_ENCODER_MAPS_WHOLE=1" > $O/ab_whole.txt 2>&1; cat $O/ab_whole.txt
