export TMPDIR=/tmp
O=gpurun_out/r6e; mkdir -p $O
bash tools/lab/build_lab.sh -DPDR_LAB_TRACE > $O/build.txt 2>&1; tail -2 $O/build.txt
timeout 600 python -m tools.lab.chain_trace > $O/chain_trace.txt 2>&1; grep -v amdgpu.ids $O/chain_trace.txt
