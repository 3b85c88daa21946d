# option ws_double_acc (one workgroup per CU, two accumulator sets) against the two-workgroup form: layers alone, then the step
for r in 1 2; do
for o in 0 1; do
  echo "== ws_double_acc=$o"
  for i in 0 1 14 13; do PDR_OPTIONS=ws_double_acc=$o python -m tools.fused_layer_bench --only $i 2>&1 | grep rpb; done
  PDR_OPTIONS=ws_double_acc=$o python -m tools.fused_layer_bench --only 0 --gath 8 --knn 2>&1 | grep rpb | sed 's/$/ knn/'
  PDR_OPTIONS=ws_double_acc=$o python -m tools.fused_layer_bench --only 1 --gath 8 --knn 2>&1 | grep rpb | sed 's/$/ knn/'
done
done
BENCH="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 5"
for r in 1 2 3; do for o in 0 1; do echo -n "step ws_double_acc=$o  "; PDR_OPTIONS=ws_double_acc=$o $BENCH 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; done; done
for r in 1 2; do for o in 0 1; do echo -n "whole ws_double_acc=$o  "; PDR_OPTIONS=ws_double_acc=$o $BENCH --neighbourhoods whole 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; done; done
