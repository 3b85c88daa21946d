B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do for v in 0 32768 131072 1048576; do echo -n "par_min_rows=$v  " | tee -a gpurun_out/c27_ab.txt; PDR_PAR_MIN_ROWS=$v $B 2>&1 | ms | tee -a gpurun_out/c27_ab.txt; done; done
echo -n "par_deep=0 " | tee -a gpurun_out/c27_ab.txt; PDR_PAR_DEEP=0 $B 2>&1 | ms | tee -a gpurun_out/c27_ab.txt
