# HBM traffic per launch of the gathered wide kernels under the plain and the XCD-local tile order (two --pmc passes each)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --warmup 3 --no-cpu-baseline --no-roofline --no-extras --steps 2"
for ord in 0 1; do for c in FETCH_SIZE WRITE_SIZE; do
  PDR_WS_XCD_ORDER=$ord rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/xcdpmc/$ord/$c -o p -- $BENCH > /dev/null 2>&1
done; done
python - <<'P'
import csv, glob, collections, os, re, json
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
res = {}
for ord_ in ("0", "1"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("%s/xcdpmc/%s/%s/**/*counter_collection.csv" % (out, ord_, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and "fused_layer_ws_kernel" in r["Kernel_Name"]:
                    k = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0]
                    acc[k][c][0] += float(r["Counter_Value"]); acc[k][c][1] += 1
    for k, v in acc.items():
        if v["FETCH_SIZE"][1] and v["WRITE_SIZE"][1]:
            mb = (2 * 1024 * v["FETCH_SIZE"][0] / v["FETCH_SIZE"][1] + 1024 * v["WRITE_SIZE"][0] / v["WRITE_SIZE"][1]) / 1e6
            res.setdefault(k, {})[ord_] = round(mb, 1)
rows = sorted(res.items(), key=lambda kv: -kv[1].get("0", 0))
for k, v in rows[:14]:
    print("%-66s plain %8s MB   xcd-local %8s MB" % (k[:66], v.get("0"), v.get("1")))
json.dump(res, open(out + "/xcd_traffic.json", "w"), indent=1)
P
rm -rf $OUT/xcdpmc
