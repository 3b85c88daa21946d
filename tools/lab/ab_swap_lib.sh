# step time of libpdr_hip.so (the product build) vs libpdr_lab.so (an experimental build), same box, alternating;
# then a subset of the GPU tests on the experimental build
L=point_diffusion_refinement_amd
cp $L/libpdr_hip.so /tmp/prod.so
BENCH="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 5"
for i in 1 2 3; do
  for which in prod lab; do
    if [ $which = prod ]; then cp /tmp/prod.so $L/libpdr_hip.so; else cp $L/libpdr_lab.so $L/libpdr_hip.so; fi
    echo -n "$which  f32 "; $BENCH 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
cp $L/libpdr_lab.so $L/libpdr_hip.so
python -m pytest tests/test_ops_gpu.py tests/test_reference_golden.py -m gpu -x -q 2>&1 | tail -2
cp /tmp/prod.so $L/libpdr_hip.so
