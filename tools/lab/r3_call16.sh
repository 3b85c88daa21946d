mkdir -p gpurun_out
python -m pytest tests/test_fused_gpu.py -m gpu -x -q 2>&1 | tail -3
PDR_SIDE_TABLES=1 python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "network or sampler or second_batch" 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do for v in 0 1; do
  echo -n "side_tables=$v  " | tee -a gpurun_out/c16_ab.txt; PDR_SIDE_TABLES=$v $B 2>&1 | ms | tee -a gpurun_out/c16_ab.txt
done; done
for v in 0 1; do echo -n "B=8 side_tables=$v  " | tee -a gpurun_out/c16_ab.txt; PDR_SIDE_TABLES=$v $B --batch 8 2>&1 | ms | tee -a gpurun_out/c16_ab.txt; done
