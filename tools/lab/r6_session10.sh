export TMPDIR=/tmp
O=gpurun_out/r6k; mkdir -p $O
AB_STEPS=60 bash tools/lab/ab_opts.sh "-" "SA0_TABLE_AHEAD=0" > $O/ab_sa0.txt 2>&1; cat $O/ab_sa0.txt
