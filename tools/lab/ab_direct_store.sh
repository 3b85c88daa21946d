L=point_diffusion_refinement_amd
for r in 1 2; do
for lib in "" "--lib $L/libpdr_lab.so"; do
  echo "== lib: ${lib:-product}"
  for i in 0 1 2 4 6; do python -m tools.fused_layer_bench --only $i $lib 2>&1 | grep rpb; done
  for i in 0 2 4; do python -m tools.fused_layer_bench --only $i --gath 8 --knn $lib 2>&1 | grep rpb | sed 's/$/ knn/'; done
done
done
