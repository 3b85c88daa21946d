import contextlib, io, numpy as np, torch
from tests.golden.det_weights import fill_deterministic
from tests.golden.tiny_config import small_fused_config
from tests.oracle_backend import oracle_ops
from tests.test_generation_gpu import _dataset, T, N
from point_diffusion_refinement_amd.pointnet2 import fused_network as FN, util
from point_diffusion_refinement_amd.pointnet2.models.pointnet2_with_pcld_condition import PointNet2CloudCondition
from point_diffusion_refinement_amd.pointnet2.reverse_sampler import GraphedReverseSampler
cuda = torch.device("cuda:0")
dh = util.calc_diffusion_hyperparams(T, 1e-4, 0.02)
net_gpu = fill_deterministic(PointNet2CloudCondition(small_fused_config()), 31).eval().to(cuda)
cond, label, gt = (t.to(cuda) for t in _dataset(0, 16))
fused = FN.FusedCloudConditionNet(net_gpu)
for flag in (True, False):
    FN.FUSE_CONDITION_BRANCH = flag
    for use_graph in (False, True):
        s = GraphedReverseSampler(fused, dh, noise='cpu', use_graph=use_graph)
        torch.manual_seed(5); a = s.sample((16, N, 3), cond, label)
        torch.manual_seed(5); b = s.sample((16, N, 3), cond, label)    # second batch through the same sampler
        util.set_device(cuda); util.set_noise_source('cpu'); torch.manual_seed(5)
        with contextlib.redirect_stdout(io.StringIO()):
            ref = util.sampling(net_gpu, (16, N, 3), dh, label=label, verbose=False, condition=cond)
        util.set_device(None)
        ea = ((a - ref).abs() / (ref.abs() + 1)).amax((1, 2)); eb = ((b - ref).abs() / (ref.abs() + 1)).amax((1, 2))
        print("fuse_cond", flag, "graph", use_graph, "first batch max", float(ea.max()), "second", float(eb.max()))
