mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py tests/test_fused_gpu.py tests/test_reference_golden.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/c4_pytest.txt
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
  for cfg in "1 1" "0 1" "1 0"; do
    set -- $cfg
    echo -n "native_embed=$1 knn_wave=$2  " | tee -a gpurun_out/c4_ab.txt
    PDR_NATIVE_EMBED=$1 PDR_KNN_WAVE=$2 $B 2>&1 | ms | tee -a gpurun_out/c4_ab.txt
  done
done
echo -n "B=8 " | tee -a gpurun_out/c4_ab.txt; $B --batch 8 2>&1 | ms | tee -a gpurun_out/c4_ab.txt
python tools/op_roofline.py > gpurun_out/c4_op_roofline.json 2> gpurun_out/c4_op_roofline.err; PDR_KNN_WAVE=0 python tools/op_roofline.py > gpurun_out/c4_op_roofline_oldknn.json 2>/dev/null
python - <<'P'
import json
for f in ("gpurun_out/c4_op_roofline.json", "gpurun_out/c4_op_roofline_oldknn.json"):
    try:
        d = json.load(open(f))
        for r in (d if isinstance(d, list) else d.get("ops", [])):
            if "knn" in str(r.get("op", r)).lower(): print(f, r)
    except Exception as e: print(f, e)
P
python -m tools.lab.step_markers gpurun_out/c4_markers.json 2>&1 | tail -32 | tee gpurun_out/c4_markers.txt
python -m tools.lab.graph_latency gpurun_out/c4_graph_latency.json > gpurun_out/c4_graph_latency.txt 2>&1; tail -12 gpurun_out/c4_graph_latency.txt
