"""Lab: per-step duration along a run of replayed steps (HIP events between the steps): does the step get faster with time?"""
import sys

import torch

import bench as BN
from point_diffusion_refinement_amd.pointnet2.configs import synthetic_batch

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
form = sys.argv[2] if len(sys.argv) > 2 else "adaptive"
sampler, _ = BN.build_sampler(dev, True, neighbourhoods=form)
x_T, cond, label = synthetic_batch(32, seed=0, device=dev)
sampler.begin((32, 2048, 3), cond, label, x_T=x_T)
sampler.begin((32, 2048, 3), cond, label, x_T=x_T)
sampler.advance(3)
torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
evs[0].record()
for i in range(n):
    sampler.advance(1)
    evs[i + 1].record()
torch.cuda.synchronize()
ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
print("first 12:", " ".join("%.2f" % m for m in ms[:12]))
for g in range(0, n, 20):
    seg = ms[g:g + 20]
    print("steps %3d-%3d: mean %.3f  min %.3f  max %.3f" % (g, g + len(seg) - 1, sum(seg) / len(seg), min(seg), max(seg)))
