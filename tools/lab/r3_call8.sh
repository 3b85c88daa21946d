mkdir -p gpurun_out
python -m pytest tests/test_fused_gpu.py tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/c8_pytest.txt
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do for v in 1 0; do
  echo -n "first_ball_main=$v  " | tee -a gpurun_out/c8_ab.txt; PDR_FIRST_BALL_MAIN=$v $B 2>&1 | ms | tee -a gpurun_out/c8_ab.txt
done; done
python -m tools.lab.step_markers gpurun_out/c8_markers.json 2>&1 | tail -32 | tee gpurun_out/c8_markers.txt
