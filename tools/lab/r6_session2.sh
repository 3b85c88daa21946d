export TMPDIR=/tmp
O=gpurun_out/r6c; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "point_chain or fp_block_second_mlp" > $O/chain_tests.txt 2>&1; tail -15 $O/chain_tests.txt
AB_STEPS=60 bash tools/lab/ab_opts.sh "-" "POINT_CHAINS=0" > $O/ab_chain.txt 2>&1; cat $O/ab_chain.txt
for q in 1 2 3 4; do echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout 300 python - <<'PY'
import json
from tools.lab.two_batches import run
for n in (1, 2):
    print(json.dumps(run(n, 40)), flush=True)
PY
done > $O/hwq.txt 2>&1; grep -v amdgpu.ids $O/hwq.txt
