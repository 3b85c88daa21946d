"""Summarise the passes of tools/profile_round.sh into the tracked round artefacts.

    python tools/profile_summary.py <gpurun_out dir> <tag>

<tag>_bench_kernel_stats.csv  rocprofv3 --kernel-trace --stats of bench.py (per kernel: calls, total, average ns)
<tag>_pmc_traffic.json        HBM bytes per launch per kernel: FETCH_SIZE (units of 1024 B; on gfx950 it tallies the
                              128-B requests of wide streaming reads at 64 B -> doubled, MI355X_MICROARCH.md, HBM
                              section) + WRITE_SIZE (as reported), from two separate passes
<tag>_mfma_util.json          per kernel: SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs) -- the counter
                              counts busy cycles of every SIMD's matrix pipe (checked: = #MFMA x 64 cycles for
                              v_mfma_f32_32x32x2_f32) -- and SQ_BUSY_CU_CYCLES-relative occupancy figures
<tag>_roofline.json           the table DESIGN.md quotes: per kernel of the step (>= 0.3 % of the time) average
                              duration, launches per step, HBM GB/s vs 8 TB/s, MFMA utilisation vs the fp32 peak,
                              which roof is nearer; plus the op-level rows of tools/op_roofline.py
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

CLOCK_HZ, SIMDS = 2.4e9, 1024
HBM_PEAK = 8.0e12


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    s = m.group(1) if m else name
    return s[:100]


def counters(path, wanted):
    """{kernel: {counter: [sum, n]}, '_dur': [sum_ns, n]}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            c = r["Counter_Name"]
            if c in wanted:
                k = short(r["Kernel_Name"])
                acc[k][c][0] += float(r["Counter_Value"])
                acc[k][c][1] += 1
                if c == wanted[0]:
                    acc[k]["_dur"][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                    acc[k]["_dur"][1] += 1
    return acc


def timeline(out, tag):
    """<tag>_timeline.json: every kernel of ONE replayed reverse step (the last complete one of the --stats pass):
    [short name, start us relative to the step's first kernel, duration us, queue id].  A step ends with the
    reverse_step (round <= 2: reverse_update) kernel; its kernels may run on several hardware queues (geometry side stream)."""
    src = glob.glob(os.path.join(out, tag + "_stats", "**", "*kernel_trace.csv"), recursive=True)
    if not src:
        return
    rows = []
    for r in csv.DictReader(open(src[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "reverse_update" in r[2] or "reverse_step" in r[2]]
    if len(ends) < 3:
        return
    lo, hi = ends[-2] + 1, ends[-1]
    # the step counter decrement (one tiny torch kernel) follows the update: include up to the next kernel start
    step = rows[lo:hi + 1]
    t0 = step[0][0]
    tl = [[n, round((a - t0) / 1e3, 2), round((b - a) / 1e3, 2), q] for a, b, n, q in step]
    span = (max(r[1] for r in step) - t0) / 1e3
    busy = {}
    for a, b, n, q in step:
        busy[q] = busy.get(q, 0.0) + (b - a) / 1e3
    json.dump({"note": "one hipGraph-replayed reverse step, rocprofv3 kernel trace; [kernel, start_us, dur_us, queue]",
               "span_us": round(span, 1), "kernels": len(tl), "busy_us_per_queue": {k: round(v, 1) for k, v in busy.items()},
               "timeline": tl}, open(os.path.join(out, tag + "_timeline.json"), "w"))


def main():
    out, tag = sys.argv[1], sys.argv[2]
    stats_src = glob.glob(os.path.join(out, tag + "_stats", "**", "*kernel_stats.csv"), recursive=True)
    stats = []
    if stats_src:
        shutil.copy(stats_src[0], os.path.join(out, tag + "_bench_kernel_stats.csv"))
        stats = list(csv.DictReader(open(stats_src[0])))
    pmc = os.path.join(out, tag + "_pmc")
    fetch = counters(os.path.join(pmc, "FETCH_SIZE"), ["FETCH_SIZE"])
    write = counters(os.path.join(pmc, "WRITE_SIZE"), ["WRITE_SIZE"])
    sq = counters(os.path.join(pmc, "SQ"), ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES",
                                            "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_VALU",
                                            "SQ_INSTS_VMEM_RD"])
    traffic = []
    for k, v in fetch.items():
        n = v["FETCH_SIZE"][1]
        fb = 2.0 * 1024.0 * v["FETCH_SIZE"][0] / n
        w = write.get(k, {}).get("WRITE_SIZE", [0.0, 0])
        wb = 1024.0 * w[0] / max(w[1], 1)
        traffic.append({"kernel": k, "launches": n, "fetch_bytes_per_launch": round(fb),
                        "write_bytes_per_launch": round(wb), "hbm_bytes_per_launch": round(fb + wb),
                        "avg_us_under_pmc": round(v["_dur"][0] / max(v["_dur"][1], 1) / 1e3, 2)})
    traffic.sort(key=lambda r: -r["hbm_bytes_per_launch"] * r["launches"])
    json.dump({"note": "FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE as reported; counter units of 1024 B; "
                       "separate rocprofv3 --pmc passes of bench.py --steps 2",
               "kernels": traffic[:60]}, open(os.path.join(out, tag + "_pmc_traffic.json"), "w"), indent=1)
    mfma = []
    for k, v in sq.items():
        n = v["SQ_VALU_MFMA_BUSY_CYCLES"][1]
        if n == 0:
            continue
        dur = v["_dur"][0] / max(v["_dur"][1], 1)           # ns
        busy = v["SQ_VALU_MFMA_BUSY_CYCLES"][0] / n
        row = {"kernel": k, "launches": n, "avg_us_under_pmc": round(dur / 1e3, 2),
               "mfma_busy_cycles_per_launch": round(busy),
               "mfma_util": round(busy / (dur * 1e-9 * CLOCK_HZ * SIMDS), 4) if dur > 0 else None}
        cu = v["SQ_BUSY_CU_CYCLES"][0] / max(v["SQ_BUSY_CU_CYCLES"][1], 1)
        if cu > 0:
            # SQ_BUSY_CU_CYCLES sums the cycles in which a CU had at least one wave: busy / (4 SIMDs x that) is the
            # matrix-pipe utilisation while a CU is occupied, and cu / (256 x duration x 2.4 GHz) the fraction of the
            # kernel's duration the average CU is occupied at all (launch ramp, tail behind the slowest workgroup).
            # Rounds 2-3 read the latter as a clock ("implied_clock_GHz"); tools/lab/clock_probe.hip and rocm-smi
            # (profiles/r4_clock_probe.txt, r4_clocks_power.txt) show the chip at 2.39-2.40 GHz under this load.
            row["mfma_util_of_busy_cu_cycles"] = round(busy / (4.0 * cu), 4)
            row["cu_occupied_frac_of_kernel_time"] = round(cu / 256.0 / dur / (CLOCK_HZ * 1e-9), 3) if dur > 0 else None
        wc = v["SQ_WAVE_CYCLES"][0] / max(v["SQ_WAVE_CYCLES"][1], 1)
        if wc > 0:
            row["issue_stalled_frac_of_wave_cycles"] = round(v["SQ_WAIT_INST_ANY"][0] / max(v["SQ_WAIT_INST_ANY"][1], 1) / wc, 3)
            row["active_frac_of_wave_cycles"] = round(v["SQ_ACTIVE_INST_ANY"][0] / max(v["SQ_ACTIVE_INST_ANY"][1], 1) / wc, 3)
            row["parked_frac_of_wave_cycles"] = round(v["SQ_WAIT_ANY"][0] / max(v["SQ_WAIT_ANY"][1], 1) / wc, 3)
        nv = v["SQ_INSTS_VALU"][0] / max(v["SQ_INSTS_VALU"][1], 1)
        nm = v["SQ_INSTS_VMEM_RD"][0] / max(v["SQ_INSTS_VMEM_RD"][1], 1)
        if nv > 0:
            # wave-level instruction counts per launch.  A VALU instruction is NOT hidden behind an fp32 MFMA: the
            # SIMD issues one or the other (tools/lab/mfma_shadow.hip: +5.5-8.5 cycles per VALU instruction next to
            # the wave's own MFMA stream, ~3 with two waves per SIMD) -- priced here at 4 cycles each
            row["valu_insts_per_launch"] = round(nv)
            row["vmem_rd_insts_per_launch"] = round(nm)
            row["valu_port_frac_of_kernel_time"] = round(nv * 4.0 / SIMDS / (dur * 1e-9 * CLOCK_HZ), 3) if dur > 0 else None
        mfma.append(row)
    mfma.sort(key=lambda r: -r["mfma_busy_cycles_per_launch"] * r["launches"])
    json.dump({"note": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 1024 SIMDs); the fp32 "
                       "MFMA peak (157.3 TF) corresponds to 1.0.  mfma_util_of_busy_cu_cycles normalises by the "
                       "cycles a CU holds at least one wave (SQ_BUSY_CU_CYCLES); cu_occupied_frac_of_kernel_time is "
                       "that occupancy as a fraction of the launch's duration (ramp + tail behind the slowest "
                       "workgroup; NOT a clock: the chip runs at 2.39-2.40 GHz under this load, "
                       "profiles/r4_clock_probe.txt).  valu_port_frac_of_kernel_time prices every VALU instruction "
                       "at 4 cycles of the SIMD's shared VALU / MFMA issue port (profiles/r4_mfma_shadow_probe.txt)",
               "kernels": mfma[:60]}, open(os.path.join(out, tag + "_mfma_util.json"), "w"), indent=1)
    # ---- roofline table
    tmap = {r["kernel"]: r for r in traffic}
    mmap = {r["kernel"]: r for r in mfma}
    rows, total = [], sum(float(r["TotalDurationNs"]) for r in stats) or 1.0
    nsteps = 20 + 3 + 2          # timed + warm-up + two first steps (bench.py: begin x2)
    serial = {}
    sp = os.path.join(out, tag + "_bench_kernel_stats_serial.csv")
    if os.path.exists(sp):
        serial = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sp))}
    for r in stats:
        share = float(r["TotalDurationNs"]) / total
        if share < 0.003:
            continue
        k = short(r["Name"])
        in_step = float(r["AverageNs"]) / 1e3
        # rates are quoted on the duration of a launch that has the chip to itself (serial pass: the two halves of
        # every block on one stream); in the step the halves overlap and a launch's wall time is longer
        avg_us = serial.get(k, in_step)
        row = {"kernel": k, "share_of_gpu_time": round(share, 4), "avg_us": round(avg_us, 2),
               "avg_us_in_step": round(in_step, 2), "launches_per_step": round(int(r["Calls"]) / nsteps, 1)}
        t = tmap.get(k)
        if t:
            gbs = t["hbm_bytes_per_launch"] / (avg_us * 1e-6) / 1e9
            row.update(hbm_MB_per_launch=round(t["hbm_bytes_per_launch"] / 1e6, 2), hbm_GBps=round(gbs, 1),
                       hbm_frac_of_8TBps=round(gbs * 1e9 / HBM_PEAK, 3))
        m = mmap.get(k)
        if m and m["mfma_util"] is not None:
            row["mfma_util"] = m["mfma_util"]
        if "hbm_frac_of_8TBps" in row or "mfma_util" in row:
            row["nearer_roof"] = "mfma" if row.get("mfma_util", 0) >= row.get("hbm_frac_of_8TBps", 0) else "hbm"
        rows.append(row)
    ops = None
    try:
        ops = json.load(open(os.path.join(out, tag + "_op_roofline.json")))
    except (OSError, ValueError):
        pass
    timeline(out, tag)
    json.dump({"note": "step kernels: rocprofv3 --kernel-trace --stats of bench.py (hipGraph replay) joined with the "
                       "PMC passes; op rows: tools/op_roofline.py (HIP events, BASELINE sizes). Peaks: HBM 8 TB/s, "
                       "fp32 MFMA = fp32 VALU = 157.3 TFLOP/s",
               "step_kernels": rows, "ops": ops}, open(os.path.join(out, tag + "_roofline.json"), "w"), indent=1)
    for r in rows[:14]:
        print("%-64s %5.1f%% %8.1f us  hbm %5s  mfma %5s" % (r["kernel"][:64], 100 * r["share_of_gpu_time"], r["avg_us"],
                                                            r.get("hbm_frac_of_8TBps", "-"), r.get("mfma_util", "-")))


if __name__ == "__main__":
    main()
