# A/B of the freshly built libpdr_hip.so against libpdr_lab.so (the previous build), same box, two rounds
L=point_diffusion_refinement_amd
python -m pytest tests/test_fused_gpu.py tests/test_generation_gpu.py -m gpu -x -q 2>&1 | tail -2
cp $L/libpdr_hip.so /tmp/new.so
for i in 1 2; do
  for which in new prev; do
    if [ $which = new ]; then cp /tmp/new.so $L/libpdr_hip.so; else cp $L/libpdr_lab.so $L/libpdr_hip.so; fi
    echo -n "$which  "; python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
cp /tmp/new.so $L/libpdr_hip.so
python -m tools.lab.layer_shapes 2>/dev/null | grep -E "^ *(2097152|1048576) " | sort -k1,1nr -k3,3n | cut -c1-130
