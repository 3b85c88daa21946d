export TMPDIR=/tmp
O=gpurun_out/r6m; mkdir -p $O
timeout 300 python -m tools.lab.sync_cost > $O/sync_cost.txt 2>&1; grep -v amdgpu.ids $O/sync_cost.txt
for a in "" "--single-stream"; do echo "bench $a"; python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline --no-extras $a 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.readline());print(d['ms_per_step'])"; done
for a in "" "--single-stream"; do echo "bench whole $a"; python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-extras --neighbourhoods whole $a 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.readline());print(d['ms_per_step'])"; done
