cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=15 > gpurun_out/r4a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4a_pytest.log
tail -5 gpurun_out/r4a_pytest.log
python bench.py > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; tail -c 600 gpurun_out/r4a_bench.json
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 5"
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for i in 1 2; do for v in 0 1; do for p in f32 split_f16; do
  echo -n "skip_fold=$v $p: " | tee -a gpurun_out/r4a_skipfold.txt; PDR_FUSED_OPTS=LAB_SKIP_FOLD=$v $B --precision $p 2>/dev/null | ms | tee -a gpurun_out/r4a_skipfold.txt
done; done; done
python -m tools.lab.layer_shapes > gpurun_out/r4a_layer_shapes.txt 2>&1; tail -3 gpurun_out/r4a_layer_shapes.txt
python -m tools.lab.step_markers gpurun_out/r4a_markers.json > gpurun_out/r4a_markers.txt 2>&1; tail -2 gpurun_out/r4a_markers.txt
