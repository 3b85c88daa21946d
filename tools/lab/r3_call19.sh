B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5 --precision split_bf16"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do for v in 128 96 64 32; do
  echo -n "split_min_cin=$v  " | tee -a gpurun_out/c19_ab.txt; PDR_SPLIT_MIN_CIN=$v $B 2>&1 | ms | tee -a gpurun_out/c19_ab.txt
done; done
