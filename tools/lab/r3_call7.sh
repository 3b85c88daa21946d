mkdir -p gpurun_out
python -m pytest tests/test_fused_gpu.py -m gpu -x -q 2>&1 | tail -12 | tee gpurun_out/c7_pytest.txt
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
  for kb in 0 16 64 128 1024; do
    echo -n "fold_tail_kb=$kb  " | tee -a gpurun_out/c7_ab.txt; PDR_FOLD_TAIL_KB=$kb $B 2>&1 | ms | tee -a gpurun_out/c7_ab.txt
  done
done
for kb in 0 128; do echo -n "B=8 fold_tail_kb=$kb  " | tee -a gpurun_out/c7_ab.txt; PDR_FOLD_TAIL_KB=$kb $B --batch 8 2>&1 | ms | tee -a gpurun_out/c7_ab.txt; done
for v in 1 0; do echo -n "first_ball_main=$v  " | tee -a gpurun_out/c7_ab.txt; PDR_FOLD_TAIL_KB=0 PDR_FIRST_BALL_MAIN=$v $B 2>&1 | ms | tee -a gpurun_out/c7_ab.txt; done
PDR_FOLD_TAIL_KB=128 python -m tools.lab.step_markers gpurun_out/c7_markers.json 2>&1 | tail -32 | tee gpurun_out/c7_markers.txt
