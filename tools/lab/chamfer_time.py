"""Chamfer K=1 kernel timing (1000 pairs of 2048^2, 8 pairs of 16384^2): python -m tools.lab.chamfer_time [lib]"""
import sys
import torch
from point_diffusion_refinement_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]
from point_diffusion_refinement_amd.pointnet2_ops import _ext
g = torch.Generator().manual_seed(0)
for nb, n in ((1000, 2048), (8, 16384), (32, 2048)):
    a = (torch.rand(nb, n, 3, generator=g) - 0.5).cuda()
    b = (torch.rand(nb, n, 3, generator=g) - 0.5).cuda()
    _ext.chamfer_nn(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        _ext.chamfer_nn(a, b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("%5d pairs of %5d^2: %8.1f us  %.0f pairs/s  %.2f T pair-evals/s" % (nb, n, ms * 1e3, nb / ms * 1e3, 2.0 * nb * n * n / ms / 1e9))
