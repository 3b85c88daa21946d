# K split among the waves of a tiny layer's workgroup (lab knob PDR_DEEP_KS = 1 / 2 / 4): tests, shapes alone, the step
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
for k in 0; do
  echo "== tests PDR_DEEP_KS=$k"
  PDR_DEEP_KS=$k timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_reference_golden.py -m gpu -x -q \
    -k "right_sized or fused_layer_matches or random_sweep or vector_staging or weighted_statistics or full_ddpm or small_config" 2>&1 | tail -3
done > $O/ks_tests.txt 2>&1
cat $O/ks_tests.txt
{
for rep in 1 2; do
  for k in 1 0; do
    echo "== rep $rep PDR_DEEP_KS=$k"
    for i in 26 21 24 22 23 17 20; do
      PDR_DEEP_KS=$k timeout 200 python -m tools.fused_layer_bench --only $i --reps 50 2>&1 | grep rpb
    done
  done
done
} > $O/ks_kernels.txt 2>&1
cat $O/ks_kernels.txt
BENCH="timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 5"
{
for i in 1 2 3; do
  for k in 1 0; do
    echo -n "PDR_DEEP_KS=$k adaptive "; PDR_DEEP_KS=$k $BENCH 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
for k in 1 0; do
  echo -n "PDR_DEEP_KS=$k whole "; PDR_DEEP_KS=$k $BENCH --neighbourhoods whole 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
} > $O/ks_step.txt 2>&1
cat $O/ks_step.txt
