"""Every pdr_fused_layer call of one cached reverse step (B = 32, DDPM config) with its shape, flags, the
kernel it dispatches to and its HIP-event time alone on the chip (block halves serialised).

    python -m tools.lab.layer_shapes > gpurun_out/layer_shapes.txt
"""
import collections

import torch

import bench
from point_diffusion_refinement_amd import _lib
from point_diffusion_refinement_amd.pointnet2 import fused_network as FN
from tools import kernel_roofline as KR


def main():
    dev = torch.device("cuda", 0)
    from point_diffusion_refinement_amd.pointnet2.configs import synthetic_batch
    sampler, _ = bench.build_sampler(dev, False, True, "f32")
    x_T, cond, label = synthetic_batch(32, bench.N_POINTS, bench.M_COND, seed=0, device=dev)
    sampler.begin((32, bench.N_POINTS, 3), cond, label, x_T=x_T)
    sampler.begin((32, bench.N_POINTS, 3), cond, label, x_T=x_T)
    lib = _lib.load()
    fn = lib.pdr_fused_layer
    rows = []

    def timed(*args):
        li = args[0]._obj
        P, Cin, Cout = args[1], args[2], args[6]
        segs = []
        for s in range(li.n_seg):
            sg = li.seg[s]
            segs.append("%d%s%s" % (sg.C, "/%d" % sg.row_div if sg.row_div > 1 else "",
                                      ("k" if sg.g_r1 else "g") if sg.gV else ""))
        flags = "".join(c for c, v in (("p", li.pre_relu), ("s", bool(li.scale)), ("r", li.post_relu), ("a", bool(li.add)),
                                       ("R", bool(li.rseg.ptr)), ("o", bool(li.oadd)), ("Y", bool(args[7])),
                                       ("S", bool(args[9]))) if v)
        plan = (_lib._c.c_int * 8)()
        lib.pdr_fused_layer_plan(args[0], P, Cin, args[3], args[4], Cout, args[7], args[8], plan)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        sym = "thin_kernel" if plan[6] and not args[9] else KR._layer_symbol(tuple(plan[:6]))
        rows.append((P, li.rows_per_batch, Cin, Cout, "+".join(segs), flags, args[10], sym,
                     e0, e1))
        return rc

    saved = sampler.net.two_streams
    sampler.net.two_streams = False
    lib.pdr_fused_layer = timed
    try:
        with torch.no_grad():
            for _ in range(2):
                sampler._step()
            rows.clear()
            for _ in range(3):
                sampler._step()
        torch.cuda.synchronize()
    finally:
        lib.pdr_fused_layer = fn
        sampler.net.two_streams = saved
    agg = collections.OrderedDict()
    for P, rpb, Cin, Cout, segs, flags, rc0, sym, e0, e1 in rows:
        k = (P, rpb, Cin, Cout, segs, flags, rc0, sym)
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1) * 1e3
    print("# P rows_per_batch Cin Cout segs flags(p pre-relu, s scale/shift, r post-relu, a add, R residual, o oadd, "
          "Y output, S stats) relu_col0 | launches/step, us each, MB algorithmic (in+out), GB/s | kernel")
    tot = 0.0
    for (P, rpb, Cin, Cout, segs, flags, rc0, sym), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        mb = 4e-6 * P * (Cin + (Cout if "Y" in flags else 0) + (Cin if "R" in flags else 0))
        print("%8d %6d %4d %4d %-18s %-9s %4d | %4.1f %7.1f %7.1f %6.0f | %s"
              % (P, rpb, Cin, Cout, segs, flags, min(rc0, 9999), n / 3, us / n, mb, mb / (us / n) * 1e3, sym.replace("fused_layer_", "")))
        tot += us / 3
    print("# total %.1f us per step in %d launches" % (tot, len(rows) // 3))


if __name__ == "__main__":
    main()
