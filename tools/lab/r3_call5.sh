mkdir -p gpurun_out
L=point_diffusion_refinement_amd
python -m pytest tests/test_fused_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/c5_pytest.txt
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
cp $L/libpdr_hip.so /tmp/new.so
for i in 1 2 3; do
  for which in new head; do
    if [ $which = new ]; then cp /tmp/new.so $L/libpdr_hip.so; else cp $L/libpdr_lab.so $L/libpdr_hip.so; fi
    echo -n "lib=$which  " | tee -a gpurun_out/c5_ab.txt; $B 2>&1 | ms | tee -a gpurun_out/c5_ab.txt
  done
done
cp /tmp/new.so $L/libpdr_hip.so
for i in 1 2; do for v in 1 0; do
  echo -n "first_ball_main=$v  " | tee -a gpurun_out/c5_ab.txt; PDR_FIRST_BALL_MAIN=$v $B 2>&1 | ms | tee -a gpurun_out/c5_ab.txt
done; done
python -m tools.lab.step_markers gpurun_out/c5_markers.json 2>&1 | tail -32 | tee gpurun_out/c5_markers.txt
