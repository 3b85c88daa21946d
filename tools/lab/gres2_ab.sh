python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "narrow or gathered_residual or optional or small_config or ddpm_config" 2>&1 | tail -4
for i in 1 2; do
  for v in 1 0; do
    echo -n "PDR_GATHER_RES_KNN=$v  "
    PDR_GATHER_RES_KNN=$v python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  done
done
