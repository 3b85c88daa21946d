# step time under runtime (not library) environment settings, same box, alternating
BENCH="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 40 --warmup 5"
ms() { tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for r in 1 2; do
  echo -n "default                           "; $BENCH 2>&1 | ms
  for kv in HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 HSA_ENABLE_INTERRUPT=0 HSA_ENABLE_SDMA=0 DEBUG_HIP_GRAPH_BATCH_SIZE=16; do
    printf "%-34s" "$kv"; env $kv $BENCH 2>&1 | ms
  done
done
