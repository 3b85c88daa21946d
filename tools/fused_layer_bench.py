"""Micro-benchmark of pdr_fused_layer at the layer shapes of the DDPM config (B=32).

    python -m tools.fused_layer_bench [--lib path/to/libpdr_lab.so]

--lib times an experimental build of the kernels (development only; the product always loads the in-tree
libpdr_hip.so).  Prints us / TFLOP/s / algorithmic GB/s per shape.
"""
import argparse
import ctypes

import torch

from point_diffusion_refinement_amd import _lib

SHAPES = [  # (rows per batch element, Cin, Cout) at B = 32
    (16384, 128, 128), (16384, 171, 128), (8192, 128, 128), (8192, 331, 128), (2048, 256, 256), (2048, 331, 256),
    (512, 512, 512), (512, 651, 256), (65536, 32, 32), (32768, 64, 64), (65536, 41, 32), (8192, 64, 128),
    (4096, 512, 512), (16384, 512, 512), (16384, 256, 256),   # steady-state probes (not network shapes)
    (512, 128, 128), (512, 64, 64), (2048, 64, 64), (2048, 128, 128), (512, 256, 256), (128, 512, 512),  # deep levels
    (64, 256, 256), (256, 256, 256), (256, 128, 128), (64, 192, 192), (512, 512, 512), (16, 512, 512),   # 21..26
    # per-source tables of the deep blocks (few rows, wide outputs): 27..32
    (16, 643, 1163), (64, 323, 1097), (64, 323, 843), (256, 323, 587), (256, 195, 585), (1024, 163, 427),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--zeros", action="store_true", help="zero operands (DVFS probe: data-dependent power)")
    ap.add_argument("--only", type=int, default=None, help="index into SHAPES")
    ap.add_argument("--first", type=int, default=0, help="skip SHAPES before this index")
    ap.add_argument("--gath", type=int, default=0, help="K > 0: the source is a gathered first conv (U[idx] + V, K "
                                                         "neighbours per query, 2048 source points per cloud)")
    ap.add_argument("--knn", action="store_true", help="with --gath: the kNN form (+ d2 r1 + w r2, no ball counts) -- "
                                                        "the source of the feature-propagation blocks' second convs")
    ap.add_argument("--split", action="store_true", help="pdr_fused_layer_f16x3 (split-f16 arithmetic) where a tile "
                                                        "variant carries it")
    args = ap.parse_args()
    if args.lib:
        _lib.LIB_PATH = args.lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B = args.batch
    st = torch.cuda.current_stream().cuda_stream
    tot = 0.0
    for rpb, Cin, Cout in (SHAPES[args.first:] if args.only is None else [SHAPES[args.only]]):
        P = B * rpb
        ldx = (Cin + 3) // 4 * 4
        X = torch.randn(P, ldx, device=dev)
        ldw = (Cout + 3) // 4 * 4
        Wt = torch.randn(Cin, ldw, device=dev) * 0.05
        bias = torch.randn(Cout, device=dev)
        scale = torch.rand(B, Cin, device=dev) + 0.5
        shift = torch.randn(B, Cin, device=dev)
        if args.zeros:
            X.zero_(), Wt.zero_(), bias.zero_(), shift.zero_()
        Y = torch.empty(P, ldw, device=dev)
        tm = lib.pdr_fused_layer_tile_rows(rpb, Cout)
        tpb = (rpb + tm - 1) // tm
        partial = torch.empty(B * tpb, Cout, 2, device=dev)
        li = _lib.LayerIn()
        li.n_seg = 1
        li.seg[0].ptr, li.seg[0].C, li.seg[0].ld, li.seg[0].row_div = X.data_ptr(), Cin, ldx, 1
        if args.gath:
            K, n_src = args.gath, 2048
            U = torch.randn(B * n_src + 1, ldx, device=dev)
            V2 = torch.randn(P // K, 2 * ldx, device=dev)
            # neighbours of a query are close in index (ball queries on FPS-ordered clouds are not, but share lines)
            idx = torch.randint(0, n_src, (P,), device=dev, dtype=torch.int32)
            cnt = torch.full((P // K,), K, device=dev, dtype=torch.int32)
            li.seg[0].ptr = U.data_ptr()
            li.seg[0].gV, li.seg[0].gV0 = V2.data_ptr(), V2.data_ptr() + 4 * ldx
            li.seg[0].g_ldv, li.seg[0].g_nsrc, li.seg[0].g_zrow = 2 * ldx, n_src, B * n_src
            li.gidx, li.gcnt, li.gK = idx.data_ptr(), cnt.data_ptr(), K
            if args.knn:
                s1, s2 = torch.rand(P, device=dev), torch.rand(P, device=dev)
                r1, r2 = torch.randn(ldx + 4, device=dev), torch.randn(ldx + 4, device=dev)
                li.gcnt, li.seg[0].gV0 = None, None
                li.gs1, li.gs2 = s1.data_ptr(), s2.data_ptr()
                li.seg[0].g_r1, li.seg[0].g_r2 = r1.data_ptr(), r2.data_ptr()
        li.scale, li.shift = scale.data_ptr(), shift.data_ptr()
        li.pre_relu, li.post_relu, li.rows_per_batch = 0, 1, rpb

        variant = lib.pdr_fused_layer_variant(rpb, Cout)
        split = args.split and variant in (4, 5, 8)
        if split:
            from point_diffusion_refinement_amd.pointnet2.fused_network import pack_f16x3
            img, nchunks = pack_f16x3(Wt, Cout, (Cin,), 64 if variant == 8 else 128)

        def call():
            if split:
                _lib.check(lib.pdr_fused_layer_f16x3(ctypes.byref(li), P, Cin, img.data_ptr(), nchunks, bias.data_ptr(),
                                                     Cout, Y.data_ptr(), ldw, partial.data_ptr(), Cout, st),
                           "fused_layer_f16x3")
            else:
                _lib.check(lib.pdr_fused_layer(ctypes.byref(li), P, Cin, Wt.data_ptr(), ldw, bias.data_ptr(), Cout,
                                               Y.data_ptr(), ldw, partial.data_ptr(), Cout, st), "fused_layer")
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / args.reps * 1e3
        tot += us
        tf = 2.0 * P * Cin * Cout / us / 1e6
        gb = 4.0 * P * (Cin + Cout) / us / 1e3
        print("rpb=%6d Cin=%4d Cout=%4d variant=%d%s: %8.1f us %6.1f TF %6.0f GB/s" %
              (rpb, Cin, Cout, variant, " split" if split else "", us, tf, gb))
    print("total %.1f us" % tot)


if __name__ == "__main__":
    main()
