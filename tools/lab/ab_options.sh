#!/bin/bash
# same-box A/B of pdr_set_option settings: ab_options.sh "name=v,name=v" ... ; "-" = defaults; 2 alternating rounds
export TMPDIR=/tmp
run() { python bench.py --steps ${AB_STEPS:-60} --warmup 5 --no-cpu-baseline --no-roofline --no-extras ${AB_ARGS} 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.readline());print(d['ms_per_step'])"; }
for r in 1 2; do
  for o in "$@"; do
    if [ "$o" = "-" ]; then echo "default            $(run)"; else echo "$o  $(PDR_OPTIONS=$o run)"; fi
  done
done
