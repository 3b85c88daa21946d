#!/bin/bash
# round-5 GPU session 1: the new kernels' unit tests, the network-level dedup tests, bench (with the trajectory leg), FPS A/B
export TMPDIR=/tmp
O=gpurun_out/r5s1
mkdir -p $O
timeout 600 python -m pytest tests/test_fused_gpu.py -q -x -k "dedup_prepare or gather_add_tiles_twin or weighted_statistics or gn_fold_skips or pooled_launch_patches or embed_select or dedup_plan or layer_tile_subset" > $O/unit.txt 2>&1
tail -25 $O/unit.txt
timeout 300 python -m pytest tests/test_ops_gpu.py -q -k "reverse_step or fps" > $O/ops.txt 2>&1
tail -5 $O/ops.txt
timeout 900 python -m pytest tests/test_fused_gpu.py -q -k "one_point or adaptive or captured_steps or ddpm_config_and_graphed or small_config" > $O/net.txt 2>&1
tail -40 $O/net.txt
timeout 600 python bench.py --steps 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -c 6000 $O/bench.json; tail -5 $O/bench.err
for T in 256 512 1024; do echo "PDR_FPS_THREADS=$T"; PDR_FPS_THREADS=$T timeout 120 python -m tools.lab.fps_time; done > $O/fps.txt 2>&1
cat $O/fps.txt
