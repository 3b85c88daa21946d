python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "optional or small_config" 2>&1 | tail -3
PDR_FUSE_SCORE_POOL=1 python -m pytest tests/test_fused_gpu.py tests/test_generation_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
  for v in 1 0; do
    echo -n "PDR_FUSE_SCORE_POOL=$v  "
    PDR_FUSE_SCORE_POOL=$v python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  done
done
