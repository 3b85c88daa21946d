import torch
from tests.golden.det_weights import fill_deterministic
from tests.golden.tiny_config import small_fused_config
from tests.test_generation_gpu import _dataset, N
from point_diffusion_refinement_amd.pointnet2 import fused_network as FN
from point_diffusion_refinement_amd.pointnet2.models.pointnet2_with_pcld_condition import PointNet2CloudCondition
cuda = torch.device("cuda:0")
net = fill_deterministic(PointNet2CloudCondition(small_fused_config()), 31).eval().to(cuda)
fused = FN.FusedCloudConditionNet(net)
cond, label, gt = (t.to(cuda) for t in _dataset(0, 16))
x = torch.randn(16, N, 3, generator=torch.Generator().manual_seed(1)).to(cuda)
ts = torch.full((16,), 7.0, device=cuda)
with torch.no_grad():
    net.reset_cond_features()
    ref = net(x, cond, ts=ts, label=label, use_retained_condition_feature=True)
    rc = {"g": net.global_feature.clone(), "uvw": [t.clone() for t in net.l_uvw],
          "enc": [t.clone() for t in net.encoder_cond_features], "dec": [t.clone() for t in net.decoder_cond_features]}
    fused.reset_cond_features()
    got = fused(x, cond, ts=ts, label=label, use_retained_condition_feature=True)
    def e(a, b): return float(((a - b).abs() / (b.abs() + 1)).max())
    print("eps", e(got, ref), "global", e(net.global_feature, rc["g"]))
    for i, (a, b) in enumerate(zip(net.l_uvw, rc["uvw"])): print("uvw", i, e(a, b))
    for i, (a, b) in enumerate(zip(net.encoder_cond_features, rc["enc"])): print("enc", i, tuple(a.shape), e(a, b))
    for i, (a, b) in enumerate(zip(net.decoder_cond_features, rc["dec"])): print("dec", i, tuple(a.shape), e(a, b))
