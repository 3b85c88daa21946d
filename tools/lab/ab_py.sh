# A/B of two versions of fused_network.py on one box: bash tools/lab/ab_py.sh <old.py> <new.py>
F=point_diffusion_refinement_amd/pointnet2/fused_network.py
for i in 1 2; do
  cp $2 $F; python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c100-160
  cp $1 $F; python bench.py --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c100-160
done
cp $2 $F
