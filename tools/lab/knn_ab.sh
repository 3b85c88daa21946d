# A/B of the kNN virtual first conv (PDR_VIRTUAL_KNN) after the tests that cover it
python -m pytest tests/test_fused_gpu.py tests/test_generation_gpu.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do
  for v in 1 0; do
    echo "PDR_VIRTUAL_KNN=$v"
    PDR_VIRTUAL_KNN=$v python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  done
done
