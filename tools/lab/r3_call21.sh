python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "split or pool" 2>&1 | tail -4
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5 --precision split_bf16"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do for v in "4,5,8" "4,5"; do echo -n "split variants=$v  " | tee -a gpurun_out/c21_ab.txt; PDR_SPLIT_VARIANTS=$v $B 2>&1 | ms | tee -a gpurun_out/c21_ab.txt; done; done
echo -n "variants 4,5,8 min_cin 64 "; PDR_SPLIT_MIN_CIN=64 $B 2>&1 | ms
