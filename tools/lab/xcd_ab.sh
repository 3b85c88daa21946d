python -m tools.lab.order_check /tmp/a.pt 2>&1 | tail -1
PDR_WS_XCD_ORDER=2 ORDER_CHECK_B=32 python -m tools.lab.order_check /tmp/b32.pt 2>&1 | tail -1
ORDER_CHECK_B=32 python -m tools.lab.order_check /tmp/a32.pt /tmp/b32.pt 2>&1 | tail -1
PDR_WS_XCD_ORDER=2 python -m tools.lab.order_check /tmp/b.pt /tmp/a.pt 2>&1 | tail -1
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 30 --warmup 5"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do for v in 0 1 2; do echo -n "xcd_order=$v  " | tee -a gpurun_out/c29_ab.txt; PDR_WS_XCD_ORDER=$v $B 2>&1 | ms | tee -a gpurun_out/c29_ab.txt; done; done
