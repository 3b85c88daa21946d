export TMPDIR=/tmp
O=gpurun_out/r6d; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "point_chain or fp_block_second_mlp" > $O/chain_tests.txt 2>&1; tail -5 $O/chain_tests.txt
timeout 600 python -m tools.lab.chain_time > $O/chain_time.txt 2>&1; grep -v amdgpu.ids $O/chain_time.txt
AB_STEPS=60 bash tools/lab/ab_opts.sh "-" "POINT_CHAINS=0" > $O/ab_chain.txt 2>&1; cat $O/ab_chain.txt
