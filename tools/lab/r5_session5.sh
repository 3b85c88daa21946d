#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5s5
mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "fps" > $O/fps_lean.txt 2>&1; tail -3 $O/fps_lean.txt
PDR_FPS_LEAN=0 timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "fps" > $O/fps_old.txt 2>&1; tail -3 $O/fps_old.txt
echo lean; python -m tools.lab.fps_time
echo old; PDR_FPS_LEAN=0 python -m tools.lab.fps_time
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_reference_golden.py -m gpu -q -k "split_f16 or dense_and_mixed or mirror" > $O/fix.txt 2>&1; tail -5 $O/fix.txt
AB_STEPS=60 bash tools/lab/r5_ab_opts.sh - -
echo "old fps:"; PDR_FPS_LEAN=0 bash tools/lab/r5_ab_opts.sh -
