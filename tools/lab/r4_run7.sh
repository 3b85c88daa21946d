cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=point_diffusion_refinement_amd
O=gpurun_out/r4i_narrow_ab.txt
cp $L/libpdr_hip.so /tmp/base.so
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 5"
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for round in 1 2; do
for v in base p0c0 p0c3 p1c3 p3c3 direct p0c3direct; do
  if [ $v = base ]; then cp /tmp/base.so $L/libpdr_hip.so; else cp $L/libpdr_lab_$v.so $L/libpdr_hip.so; fi
  echo "== $v round $round" >> $O
  if [ $round = 1 ]; then
    for sh in 8 9 10; do for g in 0 32; do python -m tools.fused_layer_bench --only $sh --reps 30 --gath $g 2>/dev/null | grep rpb >> $O; done; done
  fi
  echo -n "step ms: " >> $O; timeout 300 $B 2>/dev/null | ms >> $O
done
done
cp /tmp/base.so $L/libpdr_hip.so
( python bench.py --steps 990 --warmup 5 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/r4i_bench_long.json 2>/dev/null & BP=$!
  for i in $(seq 1 30); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power (W)" | tr '\n' ' '; echo; sleep 0.3; done > gpurun_out/r4i_clocks_power.txt; wait $BP )
cat $O
