# 128- / 64-channel chunks for the tiny per-point layers (PDR_DEEP_CHUNKS, default on) against 32-channel chunks
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_reference_golden.py -m gpu -x -q \
  -k "fused_layer_matches or random_sweep or vector_staging or thin or weighted_statistics or full_ddpm or small_config or optional_fusions or mlp_attention" 2>&1 | tail -4 > $O/deep_tests.txt
cat $O/deep_tests.txt
{
for rep in 1 2; do
  for f in 0 1; do
    echo "== rep $rep PDR_DEEP_CHUNKS=$f"
    for i in 26 21 22 23 24 25 15 16 17 18 19 20; do
      PDR_DEEP_CHUNKS=$f timeout 200 python -m tools.fused_layer_bench --only $i --reps 50 2>&1 | grep rpb
    done
  done
done
} > $O/deep_kernels.txt 2>&1
cat $O/deep_kernels.txt
BENCH="timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 5"
{
for i in 1 2 3; do
  for f in 1 0; do
    echo -n "PDR_DEEP_CHUNKS=$f adaptive "; PDR_DEEP_CHUNKS=$f $BENCH 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
for f in 1 0; do
  echo -n "PDR_DEEP_CHUNKS=$f whole "; PDR_DEEP_CHUNKS=$f $BENCH --neighbourhoods whole 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  echo -n "PDR_DEEP_CHUNKS=$f split "; PDR_DEEP_CHUNKS=$f $BENCH --precision split_f16 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
} > $O/deep_step.txt 2>&1
cat $O/deep_step.txt
