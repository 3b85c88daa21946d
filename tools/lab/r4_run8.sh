cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=point_diffusion_refinement_amd
O=gpurun_out/r4n_epi_prio.txt
cp $L/libpdr_hip.so /tmp/base.so
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 5"
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for round in 1 2; do
for v in base e2 d18 e2d12 e2d18 e2d24 e1d18; do
  if [ $v = base ]; then cp /tmp/base.so $L/libpdr_hip.so; else cp $L/libpdr_lab_$v.so $L/libpdr_hip.so; fi
  echo "== $v round $round" >> $O
  if [ $round = 1 ]; then
    for sh in 0 1 3 4 6; do python -m tools.fused_layer_bench --only $sh --reps 30 2>/dev/null | grep rpb >> $O; done
  fi
  echo -n "step ms: " >> $O; timeout 300 $B 2>/dev/null | ms >> $O
done
done
cp /tmp/base.so $L/libpdr_hip.so
echo "=== trace e2d18 shape 0" >> $O
PDR_LAB_LIB=$L/libpdr_lab_te2d18.so timeout 100 python -m tools.lab.ws_trace 0 2>&1 | grep -v amdgpu.ids | head -60 >> $O
cat $O
