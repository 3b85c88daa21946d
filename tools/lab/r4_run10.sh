cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r4q_dedup.txt
timeout 600 python -m pytest tests/test_fused_gpu.py -m gpu -q -x -k "ddpm_config_and_graphed or gather_add" 2>&1 | tail -15 >> $O
timeout 600 python -m pytest tests/test_reference_golden.py -m gpu -q -x -k "full_ddpm_config_forward" 2>&1 | tail -15 >> $O
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 5"
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for r in 1 2; do for d in 0 1; do echo -n "DEDUP=$d step ms: " >> $O; PDR_FUSED_OPTS=DEDUP=$d timeout 300 $B 2>>$O | ms >> $O; done; done
cat $O
