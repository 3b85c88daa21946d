"""HIP-event time of pdr_knn_points and pdr_ball_query at the step's shapes (B = 32):  python -m tools.lab.geometry_time"""
import os

import torch

from point_diffusion_refinement_amd import _lib

if os.environ.get("PDR_LAB_LIB"):                      # A/B against another build of the library
    _lib.LIB_PATH = os.environ["PDR_LAB_LIB"]
from point_diffusion_refinement_amd.pointnet2_ops import _ext  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for n1, n2, K in ((2048, 1024, 8), (1024, 256, 8), (256, 64, 8), (2048, 2048, 8)):
        q = (torch.rand(32, n1, 3, generator=g) * 2 - 1).to(dev)
        c = (torch.rand(32, n2, 3, generator=g) * 2 - 1).to(dev)
        for _ in range(3):
            _ext.knn_points(q, c, K)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            _ext.knn_points(q, c, K)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print("knn_points K=%d %dx%d B=32: %.1f us  %.2f TF (8 flop per pair)" % (K, n1, n2, us, 32.0 * n1 * n2 * 8 / us / 1e6))
    for m, n, r, ns in ((2048, 3072, 0.1, 32), (1024, 2048, 0.1, 32), (1024, 1024, 0.2, 32), (256, 1024, 0.2, 32),
                        (256, 256, 0.4, 32), (64, 256, 0.4, 32)):
        q = (torch.rand(32, m, 3, generator=g) * 2 - 1).to(dev)
        c = (torch.rand(32, n, 3, generator=g) * 2 - 1).to(dev)
        for _ in range(3):
            _ext.ball_query(q, c, r, ns)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            _ext.ball_query(q, c, r, ns)
        e1.record()
        torch.cuda.synchronize()
        print("ball_query %dx%d r=%.1f ns=%d B=32: %.1f us" % (m, n, r, ns, e0.elapsed_time(e1) / reps * 1e3))


if __name__ == "__main__":
    main()
