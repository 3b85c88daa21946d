python -m pytest tests/test_fused_gpu.py -m gpu -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do for v in 1 0; do echo -n "narrow3=$v  " | tee -a gpurun_out/c25_ab.txt; PDR_WS_NARROW3=$v $B 2>&1 | ms | tee -a gpurun_out/c25_ab.txt; done; done
for v in 1 0; do echo -n "split narrow3=$v  " | tee -a gpurun_out/c25_ab.txt; PDR_WS_NARROW3=$v $B --precision split_bf16 2>&1 | ms | tee -a gpurun_out/c25_ab.txt; done
