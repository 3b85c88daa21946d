cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -15 > gpurun_out/r4r_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/r4r_smoke.txt
python bench.py 2>gpurun_out/r4r_bench.err | tail -1 > gpurun_out/r4r_bench.json
cat gpurun_out/r4r_pytest_gpu.txt gpurun_out/r4r_smoke.txt; cut -c1-300 gpurun_out/r4r_bench.json
