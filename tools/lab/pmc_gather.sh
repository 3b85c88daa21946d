# SQ / TCC counters of pdr_gather_add at the level-0 shape (moments only and with the residual write)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ga
cat > /tmp/ga_run.py <<'PY'
import sys
sys.path.insert(0, sys.argv[1])
from tools.gather_add_bench import run
run(32, 2048, 2048, 32, 96, write=False, reps=3)
run(32, 3072, 2048, 32, 96, write=False, reps=3)
PY
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/sq -o p -- python /tmp/ga_run.py $GRAFT_REPO_ROOT > $OUT.sq.log 2>&1
tail -2 $OUT.sq.log
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/tcc -o p -- python /tmp/ga_run.py $GRAFT_REPO_ROOT > $OUT.tcc.log 2>&1
tail -2 $OUT.tcc.log
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/tcp -o p -- python /tmp/ga_run.py $GRAFT_REPO_ROOT > $OUT.tcp.log 2>&1
tail -2 $OUT.tcp.log
python - <<'PY'
import csv, glob, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_ga"
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gather_add" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for k, v in acc.items():
        print(os.path.basename(os.path.dirname(f)), k, ["%.4g (%.0f us)" % (a, d / 1e3) for a, d in v[-2:]])
PY
rm -rf $OUT
