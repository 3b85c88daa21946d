# plain vs XCD-local tile order of the gathered layer kernels, exact and split-f16 step, same box
BENCH="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5"
for i in 1 2 3; do
  for ord in 0 1; do
    for prec in f32 split_f16; do
      echo -n "xcd_order $ord $prec: "; PDR_WS_XCD_ORDER=$ord $BENCH --precision $prec 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
    done
  done
done
