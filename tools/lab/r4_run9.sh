cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=point_diffusion_refinement_amd
O=gpurun_out/r4p_gather_add.txt
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 5"
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for v in gaOld gaU1 gaFull; do
  echo "== $v" >> $O
  python -m tools.lab.gather_add_bench --lib $L/libpdr_lab_$v.so 2>&1 | grep -v amdgpu.ids >> $O
done
for round in 1 2; do for v in gaOld gaU1 gaFull; do
  cp $L/libpdr_lab_$v.so $L/libpdr_hip.so
  echo -n "$v step ms: " >> $O; timeout 300 $B 2>/dev/null | ms >> $O
done; done
cat $O
