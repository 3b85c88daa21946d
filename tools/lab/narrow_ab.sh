# A/B of the 128-row / 32-channel-chunk narrow tiles (PDR_NARROW_KC32)
for v in 1 0; do
  echo "PDR_NARROW_KC32=$v"
  for i in 8 9 10; do PDR_NARROW_KC32=$v python -m tools.fused_layer_bench --only $i | head -1; done
done
for i in 1 2; do
  for v in 1 0; do
    echo "PDR_NARROW_KC32=$v"
    PDR_NARROW_KC32=$v python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  done
done
python -m tools.lab.layer_shapes > gpurun_out/layer_shapes_kc32.txt 2>&1
python -m pytest tests/test_fused_gpu.py -m gpu -x -q 2>&1 | tail -4
