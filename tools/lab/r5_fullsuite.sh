#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5full
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/all.txt 2>&1
tail -30 $O/all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
