#!/bin/bash
# same-box A/B of two versions of one python file: ab_file.sh <path in repo> <alternative file> [bench args]
# prints ms/step of 3 alternating runs each (headline leg only)
export TMPDIR=/tmp
F=$1; ALT=$2; shift 2
cp $F /tmp/ab_cur.py
run() { python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-extras "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.readline());print(d['ms_per_step'])"; }
for r in 1 2 3; do
  cp /tmp/ab_cur.py $F; echo "cur  $(run "$@")"
  cp $ALT $F;           echo "alt  $(run "$@")"
done
cp /tmp/ab_cur.py $F
