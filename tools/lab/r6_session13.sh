export TMPDIR=/tmp
O=gpurun_out/r6n; mkdir -p $O
MARK_DETAIL=1 MARK_BACK_TO_BACK=6 timeout 600 python -m tools.lab.step_markers $O/markers_detail.json > $O/markers.txt 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6n/markers_detail.json'))
print(d.get('ms_per_step_with_marks'))
for m in d['marks']:
    print("%-44s %8.1f" % (m['name'], m['median_us']))
PY
