python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "split or pool" 2>&1 | tail -4
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5 --precision split_bf16"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do echo -n "split+pool  " | tee -a gpurun_out/c20_ab.txt; $B 2>&1 | ms | tee -a gpurun_out/c20_ab.txt; done
