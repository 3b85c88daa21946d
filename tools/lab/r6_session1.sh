# round 6, session 1: the chain kernel's tests, the tests touched by the ABI 0.2.0 changes, step A/B, hardware queues
export TMPDIR=/tmp
O=gpurun_out/r6b; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "point_chain or fp_block_second_mlp" > $O/chain_tests.txt 2>&1; tail -15 $O/chain_tests.txt
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_abi.py -m "gpu or not gpu" -x -q -k "reverse_step or abi or options or fps" > $O/abi_tests.txt 2>&1; tail -3 $O/abi_tests.txt
timeout 900 python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "adaptive_sampler or captured_steps_replayed or xcd_local or random_sweep" > $O/sampler_tests.txt 2>&1; tail -3 $O/sampler_tests.txt
AB_STEPS=60 bash tools/lab/ab_opts.sh "-" "POINT_CHAINS=0" > $O/ab_chain.txt 2>&1; cat $O/ab_chain.txt
AB_STEPS=40 AB_ARGS="--neighbourhoods whole" bash tools/lab/ab_opts.sh "-" "POINT_CHAINS=0" > $O/ab_chain_whole.txt 2>&1; cat $O/ab_chain_whole.txt
for q in 4 8 16; do echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout 300 python - <<'PY'
import json
from tools.lab.two_batches import run
for n in (1, 2):
    print(json.dumps(run(n, 40)), flush=True)
PY
done > $O/hwq.txt 2>&1; cat $O/hwq.txt
