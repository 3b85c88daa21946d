export TMPDIR=/tmp
O=gpurun_out/r6j; mkdir -p $O
MARK_BACK_TO_BACK=6 timeout 600 python -m tools.lab.step_markers $O/markers_loop.json > $O/markers.txt 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6j/markers_loop.json'))
print(d.get('ms_per_step_with_marks'))
prev=0
for m in d['marks']:
    print("%-40s %8.1f" % (m['name'], m['median_us']))
PY
