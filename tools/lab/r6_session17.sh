export TMPDIR=/tmp
O=gpurun_out/r6s; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "point_chain or fp_block_second_mlp" > $O/chain_tests.txt 2>&1; tail -3 $O/chain_tests.txt
timeout 600 python -m tools.lab.chain_time > $O/chain_time.txt 2>&1; grep -v amdgpu.ids $O/chain_time.txt
AB_STEPS=60 bash tools/lab/ab_opts.sh "-" "POINT_CHAINS=0" > $O/ab_chain.txt 2>&1; cat $O/ab_chain.txt
bash tools/lab/build_lab.sh -DPDR_LAB_TRACE > $O/build.txt 2>&1
timeout 600 python -m tools.lab.chain_trace > $O/chain_trace.txt 2>&1; grep -v amdgpu.ids $O/chain_trace.txt | head -60
