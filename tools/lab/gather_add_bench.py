"""pdr_gather_add alone on the chip at the shapes of a reverse step, checked against torch (Y bit for bit, moments vs
float64):  python -m tools.lab.gather_add_bench [--lib path/to/libpdr_lab.so] [--reps 30]"""
import argparse

import torch

from point_diffusion_refinement_amd import _lib

# (rows per cloud, K, Cout, source points per cloud, empty balls, kNN terms, relu_col0, written window or None)
SHAPES = [
    (65536, 32, 96, 1024, True, False, 64, None), (65536, 32, 64, 1024, True, False, 32, None),
    (32768, 32, 160, 2048, False, False, 96, None), (32768, 32, 96, 1024, True, False, 64, None),
    (8192, 32, 320, 1024, False, False, 192, None), (8192, 32, 192, 256, True, False, 128, (64, 64)),
    (16384, 8, 256, 1024, False, True, 256, None), (8192, 8, 384, 256, False, True, 384, None),
    (65536, 32, 64, 1024, True, False, 32, (0, 64)),
]


def reference(U, V2, ld, idx, cnt, s1, r1, s2, r2, B, rpb, K, Cout, n_src, relu_col0):
    P = B * rpb
    b = torch.arange(P, device=U.device) // rpb
    q = torch.arange(P, device=U.device) // K
    u = U[b * n_src + idx.long(), :Cout]
    y = u + V2[q, :Cout]
    if s1 is not None:
        y = torch.addcmul(y, s1[:, None], r1[None, :Cout])      # fma order of the kernel: + d2 r1, then + w r2
        y = torch.addcmul(y, s2[:, None], r2[None, :Cout])
    if cnt is not None:
        y = torch.where((cnt[q] <= 0)[:, None], V2[q, ld:ld + Cout], y)
    if rpb % 128:
        return y, None                                           # (per-tile moments: whole tiles only)
    f = y.double()
    f[:, relu_col0:] = f[:, relu_col0:].clamp_min(0)
    tiles = f.view(B, rpb // 128, 128, Cout)
    return y, torch.stack([tiles.sum(2), (tiles * tiles).sum(2)], -1).view(-1, Cout, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    if args.lib:
        _lib.LIB_PATH = args.lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B, st = args.batch, torch.cuda.current_stream().cuda_stream
    tot = 0.0
    for rpb, K, Cout, n_src, has_em, has_s, relu_col0, win in SHAPES:
        g = torch.Generator(device=dev).manual_seed(rpb + Cout)
        P, ld = B * rpb, (Cout + 3) // 4 * 4
        U = torch.randn(B * n_src + 1, ld, device=dev, generator=g)
        V2 = torch.randn(P // K, 2 * ld, device=dev, generator=g)
        idx = torch.randint(0, n_src, (P,), device=dev, dtype=torch.int32, generator=g)
        cnt = (torch.randint(0, 4, (P // K,), device=dev, dtype=torch.int32, generator=g) if has_em else None)
        s1 = torch.rand(P, device=dev, generator=g) if has_s else None
        s2 = torch.rand(P, device=dev, generator=g) if has_s else None
        r1 = torch.randn(ld + 4, device=dev, generator=g) if has_s else None
        r2 = torch.randn(ld + 4, device=dev, generator=g) if has_s else None
        partial = torch.empty(B * (rpb // 128), Cout, 2, device=dev)
        y0, yc = win if win else (0, -1)
        Y = torch.full((P, (yc + 3) // 4 * 4 if win else ld), float("nan"), device=dev) if (win or Cout <= 96) else None
        p = lambda t: t.data_ptr() if t is not None else None

        def call():
            _lib.check(lib.pdr_gather_add(U.data_ptr(), ld, n_src, V2.data_ptr(), V2.data_ptr() + 4 * ld if has_em else None,
                                          2 * ld, idx.data_ptr(), p(cnt), p(s1), p(r1), p(s2), p(r2), B, rpb, K, Cout,
                                          p(Y), Y.shape[1] if Y is not None else ld, partial.data_ptr(), relu_col0, y0, yc,
                                          st), "gather_add")
        call()
        torch.cuda.synchronize()
        want, wpart = reference(U, V2, ld, idx, cnt, s1, r1, s2, r2, B, rpb, K, Cout, n_src, relu_col0)
        ok_y = True
        if Y is not None:
            w = want[:, y0:y0 + (yc if win else Cout)]
            ok_y = bool(torch.equal(Y[:, :w.shape[1]], w))
        perr = float(((partial.double() - wpart).abs() / (wpart.abs() + 1.0)).max())
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
        print("rpb=%6d K=%2d Cout=%3d n_src=%4d em=%d knn=%d Y=%s: %7.1f us  %5.0f GB/s of gathered rows | Y %s, moments err %.1e"
              % (rpb, K, Cout, n_src, has_em, has_s, "-" if Y is None else ("win" if win else "all"), us,
                 4.0 * P * Cout / us / 1e3, "bit-equal" if ok_y else "DIFFERS", perr))
    print("total %.1f us" % tot)


if __name__ == "__main__":
    main()
