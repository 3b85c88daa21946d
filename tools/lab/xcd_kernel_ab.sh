# same-box A/B of the dominant gathered kernel ALONE on the chip (bench.py's live roofline) under both tile orders
for i in 1 2; do for v in 0 1; do
  echo -n "xcd_order=$v  "; PDR_WS_XCD_ORDER=$v python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel'][-34:], r['avg_launch_us'], r['frac'], r['in_step_two_streams']['avg_launch_us'])"
done; done
