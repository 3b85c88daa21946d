# round-3 call 2: tile order / grid size bit-identity + A/B, step markers, parity measurement
mkdir -p gpurun_out
python -m tools.lab.order_check /tmp/a.pt 2>&1 | tail -1
PDR_WS_XCD_ORDER=2 python -m tools.lab.order_check /tmp/b.pt /tmp/a.pt 2>&1 | tail -1 | tee gpurun_out/c2_order.txt
PDR_WS_WGS=2048 python -m tools.lab.order_check /tmp/c.pt /tmp/a.pt 2>&1 | tail -1 | tee -a gpurun_out/c2_order.txt
PDR_WS_XCD_ORDER=2 PDR_WS_WGS=2048 python -m tools.lab.order_check /tmp/d.pt /tmp/a.pt 2>&1 | tail -1 | tee -a gpurun_out/c2_order.txt
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
  for cfg in "0 512" "1 512" "2 512" "0 1024" "0 2048" "0 4096" "2 2048"; do
    set -- $cfg
    echo -n "xcd_order=$1 ws_wgs=$2  " | tee -a gpurun_out/c2_ab.txt
    PDR_WS_XCD_ORDER=$1 PDR_WS_WGS=$2 $B 2>&1 | ms | tee -a gpurun_out/c2_ab.txt
  done
done
python -m tools.lab.step_markers gpurun_out/c2_markers.json 2>&1 | tail -32 | tee gpurun_out/c2_markers.txt
PDR_PARITY_RECORD_ONLY=1 python -m pytest tests/test_reference_golden.py -m gpu -q 2>&1 | tail -3
python - <<'P'
import json
d=json.load(open('gpurun_out/parity.json'))
for r in d['records']: print('%-44s max %.2e med %.2e p999 %.2e bound %.0e %s' % (r['name'], r['max_rel'], r['median_rel'], r['p999_rel'], r['bound'], r.get('flipped_clouds','')))
P
