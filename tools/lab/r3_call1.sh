# round-3 call 1: gn_fold small form + persistent-grid size A/B, glue trace
mkdir -p gpurun_out
python -m pytest tests/test_fused_gpu.py tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/c1_pytest.txt
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do
  for cfg in "1 512" "0 512" "1 480" "1 448" "1 384"; do
    set -- $cfg
    echo -n "fold_small=$1 ws_wgs=$2  " | tee -a gpurun_out/c1_ab.txt
    PDR_GN_FOLD_SMALL=$1 PDR_WS_WGS=$2 $B 2>&1 | ms | tee -a gpurun_out/c1_ab.txt
  done
done
python -m tools.lab.glue_trace > gpurun_out/c1_glue_trace.txt 2>&1
head -60 gpurun_out/c1_glue_trace.txt
