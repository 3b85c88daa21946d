# Drawn tile walk (pdr_layer_in_t.tile_ctr) against the fixed stride, same box: tests, kernels alone on the chip,
# per-workgroup run times (lab trace build, if present), the replayed step both ways.
#   gpurun -- 'bash tools/lab/patches/r5_drawn_tile_walk_session.sh'   ->  gpurun_out/dyn_*.txt
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
L=point_diffusion_refinement_amd
timeout 900 python -m pytest tests/test_fused_gpu.py -m gpu -x -q \
  -k "drawn or gathered_residual or narrow_layers or xcd_local or fused_layer_matches" 2>&1 | tail -6 > $O/dyn_tests.txt
cat $O/dyn_tests.txt
{
for rep in 1 2; do
  for mode in "" "--ctr"; do
    echo "== rep $rep mode '$mode'"
    echo -n "knn8   "; python -m tools.fused_layer_bench --only 0 --gath 8 --knn $mode --reps 50 | head -1
    echo -n "knn8b  "; python -m tools.fused_layer_bench --only 2 --gath 8 --knn $mode --reps 50 | head -1
    echo -n "ball32 "; python -m tools.fused_layer_bench --only 0 --gath 32 $mode --reps 50 | head -1
    echo -n "plain  "; python -m tools.fused_layer_bench --only 0 $mode --reps 50 | head -1
    echo -n "p256   "; python -m tools.fused_layer_bench --only 14 $mode --reps 50 | head -1
    echo -n "p512   "; python -m tools.fused_layer_bench --only 13 $mode --reps 50 | head -1
  done
done
} > $O/dyn_kernels.txt 2>&1
cat $O/dyn_kernels.txt
if [ -f $L/libpdr_lab.so ]; then
  for mode in "" "--ctr"; do
    echo "== trace mode '$mode'"
    PDR_LAB_LIB=$L/libpdr_lab.so python -m tools.lab.ws_trace 0 --gath 8 --knn $mode 2>&1 | head -16
  done > $O/dyn_trace.txt 2>&1
  cat $O/dyn_trace.txt
fi
BENCH="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 5"
{
for i in 1 2 3; do
  for d in 1 0; do
    echo -n "PDR_WS_DYN=$d adaptive "; PDR_WS_DYN=$d $BENCH 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
for d in 1 0; do
  echo -n "PDR_WS_DYN=$d whole "; PDR_WS_DYN=$d $BENCH --neighbourhoods whole 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
} > $O/dyn_step.txt 2>&1
cat $O/dyn_step.txt
