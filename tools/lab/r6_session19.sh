export TMPDIR=/tmp
O=gpurun_out/r6u; mkdir -p $O
for o in "deep_jobs32=256,deep_jobs64=512" "deep_jobs32=2048,deep_jobs64=4096"; do echo "== $o"; PDR_OPTIONS=$o timeout 300 python -m tools.fused_layer_bench --first 27 --reps 50 2>&1 | grep rpb; done > $O/tables.txt 2>&1; cat $O/tables.txt
