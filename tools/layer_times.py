"""Per-launch timing of the fused kernels inside one cached reverse step (HIP events on the launch
stream), with the algorithmic bytes / flops of every launch.  Diagnostic for kernel work:
    python tools/layer_times.py [--batch 32]
"""
import argparse
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from point_diffusion_refinement_amd.pointnet2 import fused_network as FN  # noqa: E402
from point_diffusion_refinement_amd.pointnet2.configs import ddpm_pointnet_config, synthetic_batch  # noqa: E402
from point_diffusion_refinement_amd.pointnet2.models.pointnet2_with_pcld_condition import \
    PointNet2CloudCondition  # noqa: E402


def instrument(records):
    """Wrap the kernel entry points of fused_network with event timing + algorithmic traffic."""
    def wrap(name, fn, meta):
        def inner(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            try:
                m = meta(a, k, out)
            except Exception:   # (a diagnostic: an entry point whose signature moved on keeps its time, loses its shape)
                m = dict(P=0, Cin=0, Cout=0, bytes=0, flops=0)
            records.append((name, m, e0, e1))
            return out
        return inner

    def layer_meta(a, k, out):
        act, conv = a[0], a[1]
        src_bytes = 0
        for sg in act.segs:
            src_bytes += 4 * sg[2] * act.P // sg[4]
        if act.radd is not None:
            src_bytes += 4 * act.C * act.P
        def aligned(sg):
            t, off, C, ld = sg[:4]
            return (t.data_ptr() + 4 * off) % 16 == 0 and ld % 4 == 0 and ld >= (C + 3) // 4 * 4
        vec = all(aligned(sg) for sg in act.segs)
        if act.radd is not None:
            vec = vec and len(act.segs) == 1 and aligned(act.radd)
        desc = "+".join("%d/%d%s" % (sg[2], sg[3], "" if aligned(sg) else "!") for sg in act.segs)
        return dict(P=act.P, rpb=act.rpb, Cin=conv.Cin, Cout=conv.Cout, bytes=src_bytes + 4 * conv.Cout * act.P,
                    listed=act.dd is not None, gath=act.gidx is not None,
                    flops=2 * act.P * conv.Cin * conv.Cout, vec=vec, desc=desc + (" radd" if act.radd is not None else ""))

    FN.run_layer = wrap("fused_layer", FN.run_layer, layer_meta)
    FN.group_build = wrap("group_build", FN.group_build,
                          lambda a, k, out: dict(P=out[0].shape[0], Cin=0, Cout=out[1], bytes=4 * out[0].numel(),
                                                 flops=0))
    FN.materialize = wrap("apply_act", FN.materialize,
                          lambda a, k, out: dict(P=out.shape[0], Cin=0, Cout=out.shape[1], bytes=8 * out.numel(),
                                                 flops=0))
    split_call = FN.SplitFirstConv.__call__
    FN.SplitFirstConv.__call__ = wrap(
        "split_first", split_call,
        lambda a, k, out: dict(P=out[0].shape[0], Cin=a[1].shape[2] + 9, Cout=a[0].Cout,
                               bytes=8 * out[0].shape[0] * a[0].Cout, flops=0))
    fold = FN.Norm.fold
    FN.Norm.fold = wrap("gn_fold", fold, lambda a, k, out: dict(P=0, Cin=0, Cout=out[0].shape[1], bytes=0, flops=0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--all", action="store_true", help="every layer launch in call order (with --single-stream: alone "
                                                       "on the chip), grouped by shape at the end")
    ap.add_argument("--single-stream", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = PointNet2CloudCondition(ddpm_pointnet_config()).to(dev).eval()
    records = []
    instrument(records)
    fused = FN.FusedCloudConditionNet(net)
    if args.single_stream:
        fused.two_streams = False
    x, cond, label = synthetic_batch(args.batch, seed=0, device=dev)
    ts = torch.full((args.batch,), 999.0, device=dev)
    with torch.no_grad():
        net(x, cond, ts=ts, label=label, use_retained_condition_feature=True)
        fused.sync_condition()
        for _ in range(2):
            fused(x, cond, ts=ts - 1, label=label, use_retained_condition_feature=True)
        records.clear()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fused(x, cond, ts=ts - 2, label=label, use_retained_condition_feature=True)
        e1.record()
    torch.cuda.synchronize()
    print("eager fused step: %.2f ms" % e0.elapsed_time(e1))
    rows = []
    tot = defaultdict(float)
    for name, m, a, b in records:
        ms = a.elapsed_time(b)
        tot[name] += ms
        rows.append((ms, name, m))
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print("%-12s %7.2f ms  (%d launches)" % (k, v, sum(1 for r in rows if r[1] == k)))
    print("%-12s %9s %5s %5s %8s %8s %8s" % ("kernel", "P", "Cin", "Cout", "ms", "GB/s", "TFLOP/s"))
    for ms, name, m in sorted(rows, key=lambda r: -r[0])[:args.top]:
        print("%-12s %9d %5d %5d %8.3f %8.0f %8.1f  %s %s" % (name, m["P"], m["Cin"], m["Cout"], ms,
                                                         m["bytes"] / ms / 1e6, m["flops"] / ms / 1e9,
                                                         "" if m.get("vec", True) else "SCALAR", m.get("desc", "")))
    if args.all:
        from point_diffusion_refinement_amd import _lib
        lib = _lib.load()
        groups = defaultdict(lambda: [0, 0.0])
        print("-- every layer launch in call order: P rpb Cin Cout variant us")
        for ms, name, m in rows:
            if name != "fused_layer":
                continue
            var = lib.pdr_fused_layer_variant(m["rpb"], m["Cout"])
            tm = lib.pdr_fused_layer_tile_rows(m["rpb"], m["Cout"])
            tiles = m["P"] // m["rpb"] * ((m["rpb"] + tm - 1) // tm)
            key = (m["P"], m["rpb"], m["Cin"], m["Cout"], var, tiles, m["listed"], m["gath"])
            groups[key][0] += 1
            groups[key][1] += ms
            print("%9d %6d %5d %5d v%d tiles %5d %s%s %8.1f" % (m["P"], m["rpb"], m["Cin"], m["Cout"], var, tiles,
                                                               "L" if m["listed"] else "-", "G" if m["gath"] else "-",
                                                               ms * 1e3))
        print("-- by shape: P rpb Cin Cout variant row-tiles listed gathered: launches, total us, mean us")
        for key, (n, ms) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
            print("%9d %6d %5d %5d v%d tiles %5d %s%s  x%2d %8.1f %7.1f" % (key[0], key[1], key[2], key[3], key[4], key[5],
                                                                          "L" if key[6] else "-", "G" if key[7] else "-",
                                                                          n, ms * 1e3, ms * 1e3 / n))
    nonvec = [(ms, m) for ms, name, m in rows if name == "fused_layer" and not m.get("vec", True)]
    print("non-vector fused_layer launches: %d, %.3f ms" % (len(nonvec), sum(r[0] for r in nonvec)))


if __name__ == "__main__":
    main()
