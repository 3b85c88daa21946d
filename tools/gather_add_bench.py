"""Micro-benchmark of pdr_gather_add at the shapes of the DDPM config (B=32): reports us and GB/s
(algorithmic bytes = Y write + one U row read per position)."""
import ctypes
import sys

import torch

from point_diffusion_refinement_amd import _lib


def run(B, n_src, m, K, Cout, knn=False, reps=20, write=True, stats=True):
    lib = _lib.load()
    dev = torch.device("cuda:0")
    ld = (Cout + 3) // 4 * 4
    U = torch.randn(B, n_src, ld, device=dev)
    V = torch.randn(B, m, ld, device=dev)
    V0 = torch.randn(B, m, ld, device=dev)
    # ball-query like locality: neighbours of a query are a random subset near a centre
    centre = torch.randint(0, n_src, (B, m, 1), device=dev)
    idx = ((centre + torch.randint(-200, 200, (B, m, K), device=dev)) % n_src).int().contiguous()
    counts = torch.full((B, m), K, dtype=torch.int32, device=dev)
    s1 = torch.rand(B * m * K, device=dev) if knn else None
    r1 = torch.randn(ld, device=dev) if knn else None
    Y = torch.empty(B * m * K, ld, device=dev)
    tpb = (m * K + 127) // 128
    partial = torch.empty(B * tpb, Cout, 2, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None

    def call():
        _lib.check(lib.pdr_gather_add(p(U), ld, n_src, p(V), p(V0), ld, p(idx), p(counts), p(s1), p(r1), None, None,
                                      B, m * K, K, Cout, p(Y) if write else None, ld, p(partial) if stats else None,
                                      0, 0, -1, st), "gather_add")
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    gb = 2 * B * m * K * Cout * 4 / 1e9
    print("B=%d n_src=%d m=%d K=%d Cout=%d knn=%d write=%d stats=%d: %.1f us  %.0f GB/s" %
          (B, n_src, m, K, Cout, knn, write, stats, us, gb / us * 1e6))


if __name__ == "__main__":
    run(32, 2048, 2048, 32, 96)
    run(32, 2048, 2048, 32, 96, write=False)
    run(32, 2048, 2048, 32, 96, stats=False)
    run(32, 2048, 2048, 32, 32, stats=False)
    run(32, 2048, 1024, 32, 64)
    run(32, 1024, 256, 32, 128)
    run(32, 256, 64, 32, 256)
    run(32, 64, 256, 32, 427, knn=True)
    run(32, 256, 1024, 32, 140, knn=True)
    run(32, 1024, 2048, 32, 105, knn=True)
