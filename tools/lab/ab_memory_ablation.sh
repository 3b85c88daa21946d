# where the wide layer kernel's time goes in the memory system: lab builds that keep the instruction stream and change
# only the addresses (every tile reads the same rows / rewrites the same rows), or the store form
L=point_diffusion_refinement_amd
for lib in "" samerows sameout both nt nostore; do
  echo "== ${lib:-product}"
  arg=""; [ -n "$lib" ] && arg="--lib $L/libpdr_lab_$lib.so"
  for i in 0 1 14; do python -m tools.fused_layer_bench --only $i $arg 2>&1 | grep rpb; done
  python -m tools.fused_layer_bench --only 0 --gath 8 --knn $arg 2>&1 | grep rpb | sed 's/$/ knn/'
done
