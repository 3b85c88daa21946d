"""FPS chain timing at the BASELINE sizes (B = 32): python -m tools.lab.fps_time"""
import torch
from point_diffusion_refinement_amd.pointnet2_ops import _ext
g = torch.Generator().manual_seed(0)
for n, m in ((2048, 1024), (1024, 256), (256, 64), (64, 16), (3072, 1024)):
    x = torch.randn(32, n, 3, generator=g).cuda()
    _ext.furthest_point_sampling(x, m)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _ext.furthest_point_sampling(x, m)
    e1.record()
    torch.cuda.synchronize()
    print(n, m, "%.1f us" % (e0.elapsed_time(e1) * 100))
