export TMPDIR=/tmp
O=gpurun_out/r6h; mkdir -p $O
AB_STEPS=60 bash tools/lab/ab_options.sh "-" "deep_jobs32=1024,deep_jobs64=2048" "deep_jobs32=4096,deep_jobs64=8192" "deep_jobs32=1024,deep_jobs64=512" "deep_jobs32=256,deep_jobs64=2048" > $O/ab_deep_jobs2.txt 2>&1; cat $O/ab_deep_jobs2.txt
AB_STEPS=40 AB_ARGS="--neighbourhoods whole" bash tools/lab/ab_options.sh "-" "deep_jobs32=1024,deep_jobs64=2048" > $O/ab_deep_jobs_whole.txt 2>&1; cat $O/ab_deep_jobs_whole.txt
