# one layer workgroup per CU (the two streams' kernels co-resident on every CU) vs the default (each kernel fills the chip)
BENCH="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5"
for pct in 100 50 62 75 100 50; do
  for prec in f32 split_f16; do
    echo -n "resident $pct% $prec: "; PDR_WS_RESIDENT_PCT=$pct $BENCH --precision $prec 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
