# step time of libpdr_hip.so (new) vs libpdr_lab.so (previous build), same box, alternating; exact and split-f16
L=point_diffusion_refinement_amd
cp $L/libpdr_hip.so /tmp/new.so
BENCH="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5"
for i in 1 2 3; do
  for which in new prev; do
    if [ $which = new ]; then cp /tmp/new.so $L/libpdr_hip.so; else cp $L/libpdr_lab.so $L/libpdr_hip.so; fi
    echo -n "$which  f32 "; $BENCH 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'], end='')"
    echo -n "  split "; $BENCH --precision split_f16 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
cp /tmp/new.so $L/libpdr_hip.so
