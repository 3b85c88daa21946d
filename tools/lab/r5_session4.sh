#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5s4
mkdir -p $O
MARK_BACK_TO_BACK=6 python -m tools.lab.step_markers $O/markers_b2b.json > $O/markers_b2b.txt 2>&1
tail -34 $O/markers_b2b.txt
bash tools/lab/r5_ab_opts.sh - FPS_STREAM=0 AHEAD_DECODER_MAPS=0 SIDE_TABLES=0 FUSED_PATCH=0 TWIN_STATS=0 FUSED_PLAN=0 DEDUP_MIN_QUERIES=64
timeout 900 python -m pytest tests/test_fused_gpu.py -q -k "every_non_default_variant or one_point" > $O/variants.txt 2>&1
tail -5 $O/variants.txt
