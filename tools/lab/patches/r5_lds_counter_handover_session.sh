# LDS-counter hand-over (PDR_WS_FLAGS=1, lab) against the per-chunk barrier: tests, kernels alone, the replayed step
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
PDR_WS_FLAGS=1 timeout 900 python -m pytest tests/test_fused_gpu.py -m gpu -x -q \
  -k "gathered_residual or knn_gathered or fused_layer_matches or random_sweep or vector_staging" 2>&1 | tail -4 > $O/flg_tests.txt
cat $O/flg_tests.txt
for f in 0 1; do PDR_WS_FLAGS=$f ORDER_CHECK_B=8 timeout 300 python -m tools.lab.order_check /tmp/o$f.pt > /dev/null 2>&1; done
python -c "
import torch
a,b=torch.load('/tmp/o0.pt'),torch.load('/tmp/o1.pt')
print('B=8 four steps, barrier vs counters: identical', torch.equal(a,b), float((a-b).abs().max()))" >> $O/flg_tests.txt 2>&1
tail -1 $O/flg_tests.txt
{
for rep in 1 2; do
  for f in 0 1; do
    echo "== rep $rep PDR_WS_FLAGS=$f"
    echo -n "knn8   "; PDR_WS_FLAGS=$f timeout 200 python -m tools.fused_layer_bench --only 0 --gath 8 --knn --reps 50 2>&1 | grep rpb
    echo -n "knn8b  "; PDR_WS_FLAGS=$f timeout 200 python -m tools.fused_layer_bench --only 2 --gath 8 --knn --reps 50 2>&1 | grep rpb
    echo -n "ball32 "; PDR_WS_FLAGS=$f timeout 200 python -m tools.fused_layer_bench --only 0 --gath 32 --reps 50 2>&1 | grep rpb
    echo -n "plain  "; PDR_WS_FLAGS=$f timeout 200 python -m tools.fused_layer_bench --only 0 --reps 50 2>&1 | grep rpb
    echo -n "p256   "; PDR_WS_FLAGS=$f timeout 200 python -m tools.fused_layer_bench --only 14 --reps 50 2>&1 | grep rpb
    echo -n "p512   "; PDR_WS_FLAGS=$f timeout 200 python -m tools.fused_layer_bench --only 13 --reps 50 2>&1 | grep rpb
    echo -n "d512   "; PDR_WS_FLAGS=$f timeout 200 python -m tools.fused_layer_bench --only 6 --reps 50 2>&1 | grep rpb
  done
done
} > $O/flg_kernels.txt 2>&1
cat $O/flg_kernels.txt
BENCH="timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 5"
{
for i in 1 2 3; do
  for f in 1 0; do
    echo -n "PDR_WS_FLAGS=$f adaptive "; PDR_WS_FLAGS=$f $BENCH 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
for f in 1 0; do
  echo -n "PDR_WS_FLAGS=$f whole "; PDR_WS_FLAGS=$f $BENCH --neighbourhoods whole 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
} > $O/flg_step.txt 2>&1
cat $O/flg_step.txt
