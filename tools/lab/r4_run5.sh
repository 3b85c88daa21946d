cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m tools.lab.gn_fold_bench > gpurun_out/r4g_fold_bench.txt 2>&1; cat gpurun_out/r4g_fold_bench.txt | tail -8
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 5"
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for i in 1 2; do for t in 0 4 8; do for p in f32 split_f16; do
  echo -n "tile64=$t $p: " | tee -a gpurun_out/r4g_ms.txt; PDR_LAB_TILE64=$t timeout 300 $B --precision $p 2>gpurun_out/r4g_bench.err | ms | tee -a gpurun_out/r4g_ms.txt
done; done; done
timeout 600 python -m pytest tests/test_fused_gpu.py -m gpu -q -x -k "groupnorm_fold or fused_layer_matches or small_config or ddpm_config_and_graphed" > gpurun_out/r4g_pytest.log 2>&1; tail -3 gpurun_out/r4g_pytest.log
