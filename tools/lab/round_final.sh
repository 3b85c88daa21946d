#!/bin/bash
# final GPU session of the round: the whole GPU suite (parity records), smoke, the default bench line
export TMPDIR=/tmp
O=gpurun_out/${1:-r6}final
mkdir -p $O
rm -f gpurun_out/parity.json
timeout 2400 python -m pytest tests -m gpu -q > $O/all.txt 2>&1
tail -6 $O/all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python -c "import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'], d['value'], d['step_form'], d['one_point_neighbourhoods']['whole_evaluation'], d['trajectory']['ddpm_t1000']['weighted_mean_ms'], d['trajectory']['dense_input'], d['cpu_baseline']['value'])"
