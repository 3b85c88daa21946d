#!/bin/bash
# round-5 GPU session 3: reordered issue + fps stream + faster plan kernel: tests, markers, bench A/B
export TMPDIR=/tmp
O=gpurun_out/r5s3
mkdir -p $O
timeout 600 python -m pytest tests/test_fused_gpu.py -q -x -k "dedup_prepare or one_point or adaptive or captured_steps or ddpm_config_and_graphed or small_config or second_batch" > $O/unit.txt 2>&1
tail -15 $O/unit.txt
python -m tools.lab.step_markers $O/markers.json > $O/markers.txt 2>&1
tail -40 $O/markers.txt
timeout 600 python bench.py --steps 40 --no-cpu-baseline --no-roofline --no-extras > $O/bench_head.json 2> $O/bench.err
python -c "import json;d=json.load(open('$O/bench_head.json'));print(d['ms_per_step'], d['value'], d['step_form'])"
timeout 900 python -m pytest tests/test_fused_gpu.py -q -k "every_non_default_variant" > $O/variants.txt 2>&1
tail -15 $O/variants.txt
