mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/c9_pytest.txt
python bench.py 2>gpurun_out/c9_bench.err | tail -1 > gpurun_out/c9_bench.json; cut -c1-1500 gpurun_out/c9_bench.json
python -m tools.lab.glue_trace 2>/dev/null | head -40 > gpurun_out/c9_glue.txt; head -32 gpurun_out/c9_glue.txt
