L=point_diffusion_refinement_amd
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
cp $L/libpdr_hip.so /tmp/new.so
for i in 1 2 3; do
  for which in base radd3; do
    if [ $which = base ]; then cp /tmp/new.so $L/libpdr_hip.so; else cp $L/libpdr_lab.so $L/libpdr_hip.so; fi
    echo -n "lib=$which  " | tee -a gpurun_out/c26_ab.txt; $B 2>&1 | ms | tee -a gpurun_out/c26_ab.txt
  done
done
cp $L/libpdr_lab.so $L/libpdr_hip.so; python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "narrow or residual or sweep" 2>&1 | tail -2
cp /tmp/new.so $L/libpdr_hip.so
