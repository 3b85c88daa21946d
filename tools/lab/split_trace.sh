# per-chunk producer / consumer timeline of the split-f16 layer kernel vs the exact one (libpdr_lab.so = -DPDR_LAB_TRACE build)
mkdir -p gpurun_out
for args in "0" "0 --split" "0 --split --gath 8" "2 --split" "14 --split"; do
  echo "=== ws_trace $args"; python -m tools.lab.ws_trace $args 2>&1 | sed -n 1,3p; python -m tools.lab.ws_trace $args 2>&1 | awk '/^chunk/{f=1} f' | sed -n 1,40p
done > gpurun_out/split_trace.txt 2>&1
python tools/fused_layer_bench.py > gpurun_out/layer_bench_f32.txt 2>&1
python tools/fused_layer_bench.py --split > gpurun_out/layer_bench_split.txt 2>&1
paste gpurun_out/layer_bench_f32.txt gpurun_out/layer_bench_split.txt | cut -c1-200
