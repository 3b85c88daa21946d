mkdir -p gpurun_out
python -m tools.lab.graph_latency gpurun_out/c3_graph_latency.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c3_graph_latency.txt
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
  for v in 1 0; do
    echo -n "level_events=$v  " | tee -a gpurun_out/c3_ab.txt
    PDR_LEVEL_EVENTS=$v $B 2>&1 | ms | tee -a gpurun_out/c3_ab.txt
  done
done
echo -n "level_events=1 B=8 " | tee -a gpurun_out/c3_ab.txt; PDR_LEVEL_EVENTS=1 $B --batch 8 2>&1 | ms | tee -a gpurun_out/c3_ab.txt
echo -n "level_events=0 B=8 " | tee -a gpurun_out/c3_ab.txt; PDR_LEVEL_EVENTS=0 $B --batch 8 2>&1 | ms | tee -a gpurun_out/c3_ab.txt
python -m tools.lab.step_markers gpurun_out/c3_markers.json 2>&1 | tail -32 | tee gpurun_out/c3_markers.txt
MARK_DETAIL=1 python -m tools.lab.step_markers gpurun_out/c3_markers_detail.json 2>&1 | tail -150 > gpurun_out/c3_markers_detail.txt
python -m pytest tests/test_reference_golden.py tests/test_fused_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/c3_pytest.txt
