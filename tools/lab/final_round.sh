# end-of-round validation on the GPU box: full GPU suite, smoke, the driver's bench line, every profile artefact
TAG=${1:-r4}
python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -12 > gpurun_out/${TAG}_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/${TAG}_smoke.txt
python bench.py 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
bash tools/lab/split_stats.sh > /dev/null 2>&1 && cp gpurun_out/split_kernel_stats.csv gpurun_out/${TAG}_split_kernel_stats.csv
python -m tools.lab.split_half > gpurun_out/${TAG}_split_accuracy.json 2>/dev/null
cat gpurun_out/${TAG}_pytest_gpu.txt gpurun_out/${TAG}_smoke.txt; cut -c1-400 gpurun_out/${TAG}_bench.json
