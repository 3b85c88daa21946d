#!/bin/bash
# round-5 GPU session 2: remaining unit tests, the whole GPU suite, quick profile (kernel stats + timeline markers)
export TMPDIR=/tmp
O=gpurun_out/r5s2
mkdir -p $O
timeout 600 python -m pytest tests/test_fused_gpu.py -q -k "gather_add_tiles_twin or weighted_statistics or gn_fold_skips or pooled_launch_patches or embed_select" > $O/unit.txt 2>&1
tail -30 $O/unit.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $O/all.txt 2>&1
tail -40 $O/all.txt
PDR_PROFILE_QUICK=1 bash tools/profile_round.sh r5q > $O/profile.txt 2>&1
tail -5 $O/profile.txt
python -m tools.lab.step_markers gpurun_out/r5q_timeline_markers.json > $O/markers.txt 2>&1
tail -40 $O/markers.txt
