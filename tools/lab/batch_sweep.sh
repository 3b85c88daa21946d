# cached-step time vs batch size (graph replay, fp32)
ms() { tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for b in 1 4 8 16 32 64 128; do echo -n "B=$b  "; python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 20 --warmup 4 --batch $b 2>&1 | ms; done
