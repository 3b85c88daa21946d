# every microbenchmark shape on the exact and on the split-f16 kernels (plain and gathered sources)
mkdir -p gpurun_out
python -m tools.fused_layer_bench > gpurun_out/layer_bench_f32.txt 2>&1
python -m tools.fused_layer_bench --split > gpurun_out/layer_bench_split.txt 2>&1
python -m tools.fused_layer_bench --gath 8 > gpurun_out/layer_bench_f32_g8.txt 2>&1
python -m tools.fused_layer_bench --split --gath 8 > gpurun_out/layer_bench_split_g8.txt 2>&1
paste -d'|' gpurun_out/layer_bench_f32.txt gpurun_out/layer_bench_split.txt | grep rpb | cut -c1-170
echo; paste -d'|' gpurun_out/layer_bench_f32_g8.txt gpurun_out/layer_bench_split_g8.txt | grep rpb | cut -c1-170
