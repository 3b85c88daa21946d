"""Live roofline of the dominant hand-written kernel of the reverse step (used by bench.py).

Nothing here is hard-coded to a kernel: every C-ABI launch of `reps` eager repetitions of the step is
bracketed with HIP events recorded on the stream the kernel is launched on (torch's current stream -- the
ops read the same stream handle), launches are grouped by the kernel SYMBOL they dispatch to
(`pdr_fused_layer_plan` returns the instantiation `pdr_fused_layer` picks for a call) and the group with the
largest total time is the dominant kernel of THIS run.  For it
    achieved = (sum of algorithmic flops [or bytes] of its launches) / (sum of their durations)
with the algorithmic work of a launch stated per entry point in `_work()` (no padding, no re-reads):
    pdr_fused_layer     2 P Cin Cout flops;  4 (sum_seg C P / row_div + P Cout [+ P Cin residual]) bytes
    pdr_gather_add      table + query rows + index + written columns
    pdr_attention_pool  8 D P bytes in, 4 D P / K out
and the roof is the one that binds that kernel: max(flops / 157.3 TF, bytes / 8 TB/s).
`traffic` (HBM bytes per launch from the PMC passes of tools/profile_round.sh) is looked up BY THE SELECTED
SYMBOL in the newest profiles/r*_pmc_traffic.json; a profile that does not contain the symbol is reported as
an error string instead of a stale number.
"""
import collections
import glob
import json
import os
import re

import torch

from point_diffusion_refinement_amd import _lib

FP32_MFMA_PEAK_TFLOPS = 157.3
F16_MFMA_PEAK_TFLOPS = 2500.0
HBM_PEAK_GBS = 8000.0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# tile variants of pdr_fused_layer (csrc/fused_layer.hip pick_tile / launch tables): id -> <RT, CT, WR, WC, KC>
_VARIANT = {0: (2, 1, 4, 1, 16), 1: (2, 2, 4, 1, 16), 2: (1, 3, 4, 1, 32), 3: (1, 5, 4, 1, 32), 4: (2, 2, 2, 2, 32),
            5: (1, 2, 2, 2, 32), 6: (1, 1, 1, 4, 32), 7: (1, 1, 4, 1, 32), 8: (1, 2, 4, 1, 32)}


def _b(x):
    return "true" if x else "false"


def _layer_symbol(plan, pool=False):
    ws, vid, radd, gath, vec, split = plan
    t = ", ".join(str(v) for v in _VARIANT[vid])
    if ws:
        # (the last argument: PAIR, round 6 -- pdr_fused_layer / _pool launches are never the paired form)
        return "fused_layer_ws_kernel<%s, %s, %d, %s, %s, false>" % (t, _b(radd), int(gath), _b(split), _b(pool))
    return "fused_layer_kernel<%s, %s, %s, %s, %s>" % (t, _b(radd), _b(vec), _b(gath), _b(pool))


def _work(name, args, lib):
    """-> (kernel symbol, algorithmic flops, algorithmic bytes) of one C-ABI call."""
    if name == "pdr_fused_layer":
        li = args[0]._obj
        P, Cin, Cout, ldw, ldy = args[1], args[2], args[6], args[4], args[8]
        plan = (_lib._c.c_int * 8)()
        lib.pdr_fused_layer_plan(args[0], P, Cin, args[3], ldw, Cout, args[7], ldy, plan)
        byt = 4.0 * P * Cout if args[7] else 0.0
        for s in range(li.n_seg):
            sg = li.seg[s]
            if sg.gV:                                   # gathered source: table + per-query rows + index
                byt += 4.0 * sg.C * (P // max(li.gK, 1)) + 4.0 * P
            else:
                byt += 4.0 * sg.C * (P // sg.row_div)
        if li.rseg.ptr:
            byt += 4.0 * P * Cin
        if plan[6] and not args[9]:                      # <= 4 input channels, no prologue, no statistics
            return "fused_layer_thin_kernel", 2.0 * P * Cin * Cout, byt
        return _layer_symbol(tuple(plan[:6])), 2.0 * P * Cin * Cout, byt
    if name == "pdr_fused_layer_f16x3":
        li = args[0]._obj
        P, Cin, Cout = args[1], args[2], args[6]
        vid = lib.pdr_fused_layer_variant(li.rows_per_batch, Cout)
        gath = max([0] + [(2 if li.seg[s].g_r1 else 1) for s in range(li.n_seg) if li.seg[s].gV])
        byt = 4.0 * P * Cout + sum(4.0 * li.seg[s].C * (P // li.seg[s].row_div) for s in range(li.n_seg))
        if li.rseg.ptr:
            byt += 4.0 * P * Cin
        return _layer_symbol((1, vid, bool(li.rseg.ptr), gath, 1, 1)), 2.0 * P * Cin * Cout, byt
    if name == "pdr_fused_layer_pool":
        # scores = layer(in) stay in the accumulators; read: layer input + value rows, written: pooled rows
        li = args[0]._obj
        P, Cin, D, K = args[1], args[2], args[6], args[13]
        vid = lib.pdr_fused_layer_variant(li.rows_per_batch, D)
        tm = lib.pdr_fused_layer_tile_rows(li.rows_per_batch, D)
        ws = int(li.rows_per_batch % tm == 0 and vid not in (3, 6))
        byt = 4.0 * P * (Cin + D) + 4.0 * (P // K) * D
        return _layer_symbol((ws, vid, False, 0, 1, 0), pool=True), 2.0 * P * Cin * D, byt
    if name == "pdr_gather_add":
        ldu, n_src, B, rpb, K, Cout, Y, ycols = args[1], args[2], args[12], args[13], args[14], args[15], args[16], args[21]
        P = B * rpb
        byt = 4.0 * (B * n_src * Cout + (P // K) * Cout + P) + (4.0 * P * (ycols if ycols > 0 else Cout) if Y else 0.0)
        lpr = 16 if Cout <= 64 else (32 if Cout <= 128 else 64)
        return "gather_add_kernel<%d>" % lpr, 2.0 * P * Cout, byt
    if name == "pdr_attention_pool":
        B, npoint, K, D = args[8], args[9], args[10], args[11]
        P = B * npoint * K
        return "attention_pool_*", 6.0 * P * D, 8.0 * P * D + 4.0 * B * npoint * D
    if name == "pdr_gn_fold":
        # per-tile partial moments in, (B, C) scale / shift out; no arithmetic worth counting: a latency-bound launch
        tpb0, C0, tpb1, C1, B = args[2], args[3], args[7], args[8], args[10]
        C = C0 + (C1 if args[5] else 0)
        return "gn_fold_kernel", 0.0, 8.0 * B * (tpb0 * C0 + (tpb1 * C1 if args[5] else 0)) + 8.0 * B * C
    if name == "pdr_furthest_point_sampling":
        # m - 1 rounds of N distance updates (8 flops per pair); 12 N + 4 m bytes per cloud.  VALU-side
        # work on 32 of 256 CUs behind a chain of m - 1 dependent rounds: the floor is the round latency, not a roof
        B, N, m = args[1], args[2], args[3]
        return "fps_*_kernel N=%d m=%d" % (N, m), 8.0 * B * N * max(m - 1, 0), B * (12.0 * N + 4.0 * m), "valu"
    if name == "pdr_ball_query":
        B, n, m, ns = args[2], args[3], args[4], args[6]
        return "ball_query_kernel", 8.0 * B * n * m, B * (12.0 * (n + m) + 4.0 * m * ns + 4.0 * m), "valu"
    if name in ("pdr_knn_group", "pdr_knn_points"):
        B, n1, n2, K = args[2], args[3], args[4], args[5]
        return "knn (K=%d)" % K, 8.0 * B * n1 * n2, B * (12.0 * (n1 + n2) + 12.0 * n1 * K), "valu"
    if name == "pdr_embed_linear":
        B, K, N = args[8], args[9], args[10]
        return "embed_linear_kernel", 2.0 * B * K * N, 4.0 * (K * N + B * K + B * N)
    if name == "pdr_reverse_step":
        return "reverse_step_kernel", 0.0, 36.0 * args[12]
    if name == "pdr_pad_rows":
        return "pad_rows_kernel", 0.0, 4.0 * args[1] * (args[2] + args[4])
    if name == "pdr_gather_rows":
        B, C, m = args[2], args[4], args[5]
        return "gather_rows_cl_kernel", 0.0, B * m * (8.0 * C + 4.0)
    if name == "pdr_apply_act":
        return "apply_act_kernel", 0.0, 8.0 * args[1] * args[2]
    return name.replace("pdr_", "") + " (C ABI)", 0.0, 0.0


_TIMED = ("pdr_fused_layer", "pdr_fused_layer_f16x3", "pdr_fused_layer_pool", "pdr_gather_add", "pdr_attention_pool", "pdr_gn_fold", "pdr_apply_act", "pdr_gather_rows",
          "pdr_furthest_point_sampling", "pdr_ball_query", "pdr_knn_points", "pdr_group_build", "pdr_knn_build",
          "pdr_knn_weights", "pdr_pad_rows", "pdr_knn_group", "pdr_embed_linear", "pdr_reverse_step")


def latest_traffic_file():
    files = glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))

    def rnd(p):
        m = re.search(r"r(\d+)_pmc_traffic", os.path.basename(p))
        return int(m.group(1)) if m else -1
    return max(files, key=rnd) if files else None


def measured_traffic(symbol):
    """(HBM bytes per launch, source file) of `symbol` from the newest committed PMC passes (FETCH_SIZE doubled
    per the gfx950 correction + WRITE_SIZE, separate rocprofv3 runs of this bench)."""
    path = latest_traffic_file()
    if path is None:
        raise LookupError("no profiles/r*_pmc_traffic.json")
    key = symbol.replace(" ", "")
    stem = key[:-1] if key.endswith(">") else key       # a profile of a build with more (trailing) template arguments
    kernels = json.load(open(path))["kernels"]
    for exact in (True, False):
        for k in kernels:
            name = k["kernel"].replace(" ", "")
            if (key in name) if exact else (stem in name and name[name.index(stem) + len(stem)] in ",>"):
                return k["hbm_bytes_per_launch"], os.path.basename(path)
    raise LookupError("%s has no entry for %s -- re-run tools/profile_round.sh" % (os.path.basename(path), symbol))


def step_kernel_table(sampler, reps=3):
    """HIP-event timing of every C-ABI launch of `reps` eager steps -> {symbol: [n, ms, flops, bytes]}.
    The blocks' two halves run on ONE stream for the measurement, so an event bracket times a kernel that has the
    chip to itself (what a kernel roofline is about).  How long the same launch takes INSIDE the two-stream graph
    replay is not something eager event brackets can show (round 3 reported such a figure; the rocprofv3 kernel
    trace of the replayed step disagreed): that number is profiles/*_bench_kernel_stats.csv's."""
    lib = _lib.load()
    net = sampler.net
    saved_par = getattr(net, "two_streams", None)
    if saved_par is not None:
        net.two_streams = False
    # every launch over ALL its tiles: the algorithmic flops / bytes of _work() are those of whole launches, and a
    # kernel's roofline is a property of the kernel, not of how many of its tiles an input happens to need (the
    # headline step evaluates one-point neighbourhoods once: fused_network.DEDUP, a device-side tile subset)
    saved_mode, sampler._mode = sampler._mode, 'whole'
    saved_nb, sampler.neighbourhoods = sampler.neighbourhoods, 'whole'     # (no probe launches in the timed steps)
    records = []
    installed = []
    for name in _TIMED:
        if name not in _lib.SIGNATURES:
            continue
        fn = getattr(lib, name)

        def make(name, fn):
            def timed(*args):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = fn(*args)
                e1.record()
                w = _work(name, args, lib)
                records.append((e0, e1) + tuple(w[:3]) + (w[3] if len(w) > 3 else "mfma",))
                return rc
            return timed
        setattr(lib, name, make(name, fn))
        installed.append((name, fn))
    try:
        with torch.no_grad():
            for _ in range(reps):
                sampler._step()           # eager: the graph is not involved
        torch.cuda.synchronize()
    finally:
        sampler._mode, sampler.neighbourhoods = saved_mode, saved_nb
        if saved_par is not None:
            net.two_streams = saved_par
        for name, fn in installed:
            setattr(lib, name, fn)        # back to the CDLL's own (typed) function object
    table = collections.OrderedDict()
    for e0, e1, sym, fl, by, kind in records:
        row = table.setdefault(sym, [0, 0.0, 0.0, 0.0, kind])
        row[0] += 1
        row[1] += e0.elapsed_time(e1)
        row[2] += fl
        row[3] += by
    return table


def executed_gflop_per_step(sampler, mode):
    """GEMM work ONE step of form `mode` ('once' / 'whole') executes on the sampler's current x_t, in GFLOP: the sum over
    the step's layer launches (pdr_fused_layer*, the pooled score convs, the thin coordinate tables, pdr_embed_linear)
    of 2 rows Cin Cout with rows = 128 x the launch's tile count where it walks a tile subset (read back from the
    device after the step), else all of its rows.  Not counted: the adds of pdr_gather_add (P Cout each), statistics,
    pooling.  The REFERENCE's composition of the same step is 26.35 GFLOP per cloud: the difference is the
    split first conv (per-point tables instead of a GEMM over the grouped tensor) and, for 'once', the neighbourhoods
    evaluated once."""
    from point_diffusion_refinement_amd.pointnet2 import fused_network as FN
    lib = _lib.load()
    calls, plans = [], {}
    init = FN.Dedup.__init__

    def rec_plan(self, *a, **k):
        init(self, *a, **k)
        plans[self.n_tiles.data_ptr()] = self.n_tiles
    installed = []

    def hook(name, fn, P_i, Cin_i, Cout_i):
        def wrapped(*args):
            li = args[0]._obj if hasattr(args[0], "_obj") else None
            ptr = int(li.n_tiles or 0) if (li is not None and li.tile_list) else 0
            calls.append((int(args[P_i]), int(args[Cin_i]), int(args[Cout_i]), ptr))
            return fn(*args)
        setattr(lib, name, wrapped)
        installed.append((name, fn))
    saved_mode = sampler._mode
    FN.Dedup.__init__ = rec_plan
    try:
        for name, idx in (("pdr_fused_layer", (1, 2, 6)), ("pdr_fused_layer_f16x3", (1, 2, 6)),
                          ("pdr_fused_layer_pool", (1, 2, 6)), ("pdr_fused_layer_pool_f16x3", (1, 2, 6)),
                          ("pdr_embed_linear", (8, 9, 10))):
            hook(name, getattr(lib, name), *idx)
        sampler._mode = mode
        state = [v.clone() for v in (sampler._x, sampler._t, sampler._ts, sampler._rng, sampler._probe)]
        with torch.no_grad():
            sampler._step()
        torch.cuda.synchronize()
        for v, w in zip((sampler._x, sampler._t, sampler._ts, sampler._rng, sampler._probe), state):
            v.copy_(w)
    finally:
        FN.Dedup.__init__ = init
        sampler._mode = saved_mode
        for name, fn in installed:
            setattr(lib, name, fn)
    total = 0.0
    for P, Cin, Cout, ptr in calls:
        rows = 128 * int(plans[ptr]) if ptr else P
        total += 2.0 * rows * Cin * Cout
    return total / 1e9


def _roof(flops, byt, ms, symbol, kind="mfma"):
    # split mode: three f16 MFMAs per algorithmic product -> a third of the dense f16 peak
    args = symbol[symbol.find("<") + 1:symbol.rfind(">")].split(", ") if "<" in symbol else []
    split = symbol.startswith("fused_layer_ws_kernel") and len(args) >= 8 and args[7] == "true"
    peak_tf = F16_MFMA_PEAK_TFLOPS / 3.0 if split else FP32_MFMA_PEAK_TFLOPS
    t_mfma = flops / (peak_tf * 1e12)
    t_hbm = byt / (HBM_PEAK_GBS * 1e9)
    if t_mfma >= t_hbm and flops > 0:
        ach = flops / (ms * 1e-3) / 1e12
        # kind "valu": fp32 vector work (157.3 TF is the fp32 VALU peak as well as the fp32 MFMA peak)
        return {"bound": kind, "achieved": round(ach, 2), "peak": round(peak_tf, 1), "unit": "TFLOP/s",
                "frac": round(ach / peak_tf, 4)}
    ach = byt / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4)}


def dominant_kernel_roofline(sampler, reps=3):
    table = step_kernel_table(sampler, reps)
    ranked = sorted(table.items(), key=lambda kv: -kv[1][1])
    sym, (n, ms, fl, by, kind) = ranked[0]
    out = _roof(fl, by, ms, sym, kind)
    try:
        out["traffic"], out["traffic_source"] = measured_traffic(sym)
    except (LookupError, OSError, KeyError, ValueError) as e:
        out["traffic"], out["traffic_error"] = None, str(e)
    total = sum(v[1] for v in table.values())
    out.update({
        "kernel": sym, "launches_per_step": n // reps, "avg_launch_us": round(ms / n * 1e3, 2),
        "avg_gflop_per_launch": round(fl / n / 1e9, 3), "avg_algorithmic_MB_per_launch": round(by / n / 1e6, 2),
        "share_of_step_kernel_time": round(ms / total, 4),
        "note": "dominant = largest HIP-event time among the C-ABI launches of %d eager steps with the blocks' halves "
                "serialised on one stream (a kernel alone on the chip; agrees with profiles/*_bench_kernel_stats_serial"
                ".csv); events include ~3 us of eager launch latency per call; measured with every neighbourhood "
                "evaluated (whole launches: the shares are those of the whole evaluation, not of the headline step, "
                "whose per-neighbour launches walk a tile subset)" % reps,
        "next": [{"kernel": s, "share": round(v[1] / total, 4), "avg_launch_us": round(v[1] / v[0] * 1e3, 2),
                  **_roof(v[2], v[3], v[1], s, v[4])} for s, v in ranked[1:6]],
    })
    return out
