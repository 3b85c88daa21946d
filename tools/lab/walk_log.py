"""Lab: which launches of one cached step walk their tiles backwards (ZIGZAG_WALK), for the big activations."""
import torch

import bench as BN
from point_diffusion_refinement_amd.pointnet2 import fused_network as FN
from point_diffusion_refinement_amd.pointnet2.configs import synthetic_batch

dev = torch.device("cuda:0")
sampler, _ = BN.build_sampler(dev, use_graph=False)
x_T, cond, label = synthetic_batch(32, seed=0, device=dev)
sampler.begin((32, 2048, 3), cond, label, x_T=x_T)
orig = FN.Act.walk
log = []


def walk(self):
    r = orig(self)
    if self.P >= 262144:
        segs = [(sg[2], "g" if len(sg) > 5 and sg[5] is not None else "p", sg[0].data_ptr() in FN._WALK) for sg in self.segs]
        log.append((self.P, self.C, segs, None if self.radd is None else (self.radd[2], self.radd[0].data_ptr() in FN._WALK),
                    self.dd is not None, r))
    return r


FN.Act.walk = walk
sampler.advance(1)
torch.cuda.synchronize()
for row in log:
    print(row)
