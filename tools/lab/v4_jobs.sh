export TMPDIR=/tmp
BENCH="timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 5"
for i in 1 2 3; do
  for j in 128 256 512; do
    echo -n "PDR_DEEP_V4_JOBS=$j "; PDR_DEEP_V4_JOBS=$j $BENCH 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
for j in 128 256 512; do for i in 18 19 22; do echo -n "jobs $j: "; PDR_DEEP_V4_JOBS=$j python -m tools.fused_layer_bench --only $i --reps 50 2>&1 | grep rpb; done; done
