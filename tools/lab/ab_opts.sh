#!/bin/bash
# same-box A/B of PDR_FUSED_OPTS settings: ab_opts.sh "OPTS1" "OPTS2" ... ; "-" = defaults; 2 alternating rounds
export TMPDIR=/tmp
run() { python bench.py --steps ${AB_STEPS:-60} --warmup 5 --no-cpu-baseline --no-roofline --no-extras ${AB_ARGS} 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.readline());print(d['ms_per_step'])"; }
for r in 1 2; do
  for o in "$@"; do
    if [ "$o" = "-" ]; then echo "default            $(run)"; else echo "$o  $(PDR_FUSED_OPTS=$o run)"; fi
  done
done
