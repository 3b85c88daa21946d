# A/B of the freshly built libpdr_hip.so against a previous build kept as libpdr_lab.so (same box, two rounds)
python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "random_sweep or ddpm_config or matches_torch" 2>&1 | tail -2
L=point_diffusion_refinement_amd
for i in 1 2; do
  echo -n "new  "; python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  cp $L/libpdr_hip.so /tmp/new.so; cp $L/libpdr_lab.so $L/libpdr_hip.so
  echo -n "prev "; python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  cp /tmp/new.so $L/libpdr_hip.so
done
python -m tools.fused_layer_bench --first 8 --reps 30 | grep -E "variant=[01]|rpb=   512 Cin=  64"
