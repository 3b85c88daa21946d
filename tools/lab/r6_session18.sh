export TMPDIR=/tmp
O=gpurun_out/r6t; mkdir -p $O
timeout 1500 python -m pytest tests/test_fused_gpu.py tests/test_reference_golden.py -m gpu -x -q -k "gather_add or grouping or split_first or dedup or full_ddpm_config_forward or mlp_attention" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for i in 1 2 3; do python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.readline());print(d['ms_per_step'])"; done
for i in 1 2; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-extras --neighbourhoods whole 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.readline());print('whole', d['ms_per_step'])"; done
