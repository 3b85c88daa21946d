export TMPDIR=/tmp
O=gpurun_out/r6r; mkdir -p $O
timeout 1500 python -m pytest tests/test_fused_gpu.py -m gpu -x -q -k "every_non_default_variant or adaptive_sampler or one_point" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for i in 1 2 3; do python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.readline());print(d['ms_per_step'])"; done
