import torch
from tests.golden.det_weights import fill_deterministic
from tests.golden.tiny_config import small_fused_config
from tests.test_generation_gpu import _dataset, T, N
from point_diffusion_refinement_amd.pointnet2 import fused_network as FN, util
from point_diffusion_refinement_amd.pointnet2.models.pointnet2_with_pcld_condition import PointNet2CloudCondition
from point_diffusion_refinement_amd.pointnet2.reverse_sampler import GraphedReverseSampler
cuda = torch.device("cuda:0")
dh = util.calc_diffusion_hyperparams(T, 1e-4, 0.02)
net = fill_deterministic(PointNet2CloudCondition(small_fused_config()), 31).eval().to(cuda)
fused = FN.FusedCloudConditionNet(net)
cond, label, gt = (t.to(cuda) for t in _dataset(0, 16))
traj = {}
for flag in (True, False):
    FN.FUSE_CONDITION_BRANCH = flag
    s = GraphedReverseSampler(fused, dh, noise='cpu', use_graph=False)
    torch.manual_seed(5)
    s.begin((16, N, 3), cond, label)
    xs = [s._x.clone()]
    while s.remaining > 0:
        s.advance(1)
        xs.append(s._x.clone())
    traj[flag] = xs
    s.finish() if s.remaining else fused.reset_cond_features()
for i, (a, b) in enumerate(zip(traj[True], traj[False])):
    print("after step", i + 1, float(((a - b).abs() / (b.abs() + 1)).max()))
