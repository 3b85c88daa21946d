"""Lab: a chain of dependent wide layers (layer -> GroupNorm fold -> layer -> ...) over the whole batch on one stream
against the same chain over the two HALVES of the batch on two streams, both as hipGraphs.

    python -m tools.lab.half_batch [--rpb 16384] [--c 128] [--layers 4] [--batch 32]

GroupNorm is per cloud, so the halves are independent chains; the question is whether the second half's launches fill
the first half's launch tails and fold gaps (the feature-propagation blocks run alone on the chip, one launch at a time).
"""
import argparse
import ctypes

import torch

from point_diffusion_refinement_amd import _lib


def build(lib, B, rpb, C, layers, dev):
    P = B * rpb
    X0 = torch.randn(P, C, device=dev)
    acts = [X0] + [torch.empty(P, C, device=dev) for _ in range(layers)]
    Ws = [torch.randn(C, C, device=dev) * 0.05 for _ in range(layers)]
    bias = [torch.randn(C, device=dev) for _ in range(layers)]
    gamma = [torch.rand(C, device=dev) + 0.5 for _ in range(layers)]
    beta = [torch.randn(C, device=dev) for _ in range(layers)]
    tm = lib.pdr_fused_layer_tile_rows(rpb, C)
    tpb = (rpb + tm - 1) // tm
    partial = [torch.empty(B * tpb, C, 2, device=dev) for _ in range(layers)]
    scale = [torch.ones(B, C, device=dev)] + [torch.empty(B, C, device=dev) for _ in range(layers)]
    shift = [torch.zeros(B, C, device=dev)] + [torch.empty(B, C, device=dev) for _ in range(layers)]
    keep = (acts, Ws, bias, gamma, beta, partial, scale, shift)

    def run(b0, nb, st, zigzag=False):
        """the chain over clouds b0 .. b0 + nb - 1 on stream st (zigzag: every other layer walks its tiles backwards)"""
        for l in range(layers):
            li = _lib.LayerIn()
            li.n_seg = 1
            x = acts[l][b0 * rpb:]
            li.seg[0].ptr, li.seg[0].C, li.seg[0].ld, li.seg[0].row_div = x.data_ptr(), C, C, 1
            li.scale, li.shift = scale[l][b0:].data_ptr(), shift[l][b0:].data_ptr()
            li.pre_relu, li.post_relu, li.rows_per_batch = 0, 1 if l else 0, rpb
            li.walk_reverse = (l & 1) if zigzag else 0
            y, p = acts[l + 1][b0 * rpb:], partial[l][b0 * tpb:]
            _lib.check(lib.pdr_fused_layer(ctypes.byref(li), nb * rpb, C, Ws[l].data_ptr(), C, bias[l].data_ptr(), C,
                                           y.data_ptr(), C, p.data_ptr(), C, st), "fused_layer")
            _lib.check(lib.pdr_gn_fold(p.data_ptr(), C, tpb, C, 1.0, None, 0, 0, 0, 0.0, nb, C, 32, float(rpb), 1e-5,
                                       gamma[l].data_ptr(), beta[l].data_ptr(), scale[l + 1][b0:].data_ptr(),
                                       shift[l + 1][b0:].data_ptr(), None, 0, None, 0, st), "gn_fold")
    return run, keep


def timed(g, reps):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rpb", type=int, default=16384)
    ap.add_argument("--c", type=int, default=128)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--zigzag", action="store_true", help="instead of parts: the whole batch, every other layer walking "
                    "its row tiles from the last to the first (pdr_layer_in_t.walk_reverse)")
    ap.add_argument("--serial", action="store_true", help="the parts one after the other on ONE stream (each part's "
                    "activations may then stay in the 256-MB memory-side cache from one layer to the next)")
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B = args.batch
    run, keep = build(lib, B, args.rpb, args.c, args.layers, dev)
    side = [torch.cuda.Stream() for _ in range(args.parts - 1)]
    cap = torch.cuda.Stream()

    def whole():
        run(0, B, torch.cuda.current_stream().cuda_stream)

    def parts():
        main_s = torch.cuda.current_stream()
        nb = B // args.parts
        if args.zigzag:
            run(0, B, main_s.cuda_stream, zigzag=True)
            return
        if args.serial:
            for i in range(args.parts):
                run(i * nb, nb, main_s.cuda_stream)
            return
        for i, s in enumerate(side):
            s.wait_stream(main_s)
            with torch.cuda.stream(s):
                run((i + 1) * nb, nb, s.cuda_stream)
        run(0, nb, main_s.cuda_stream)
        for s in side:
            main_s.wait_stream(s)

    out = {}
    ref = None
    for name, fn in (("whole", whole), ("parts", parts)):
        with torch.cuda.stream(cap):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            fn()
        out[name] = timed(g, args.reps)
        y = keep[0][-1].clone()
        if ref is None:
            ref = y
        else:
            print("same bits:", bool(torch.equal(ref, y)))
    print("rpb=%d C=%d layers=%d B=%d: whole %.1f us, %d parts on %d streams %.1f us (%.3f x)" %
          (args.rpb, args.c, args.layers, B, out["whole"], args.parts, 1 if args.serial else args.parts, out["parts"],
           out["whole"] / out["parts"]))


if __name__ == "__main__":
    main()
