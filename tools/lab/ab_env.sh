# A/B of one environment knob on one box: bash tools/lab/ab_env.sh VAR v1 v2 ...   (two rounds)
VAR=$1; shift
for i in 1 2; do for v in "$@"; do
  echo -n "$VAR=$v  "
  env $VAR=$v python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done; done
