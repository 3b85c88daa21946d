"""Child process of tests/test_fused_gpu.py::test_ddpm_forward_with_every_non_default_variant: ONE first (uncached)
and ONE cached forward of the fused network on the shipped DDPM architecture (B = 2, seeded inputs and weights) under
whatever kernel-selection options (PDR_OPTIONS="name=value,..." -> pdr_set_option, applied by the binding) and evaluation variants
(PDR_FUSED_OPTS, point_diffusion_refinement_amd/pointnet2/fused_network.py) the environment carries; writes both eps.
    PDR_OPTIONS=fused_ws=0 python -m tools.variant_check /tmp/eps.pt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from point_diffusion_refinement_amd.pointnet2.configs import ddpm_pointnet_config, synthetic_batch
    from point_diffusion_refinement_amd.pointnet2.fused_network import FusedCloudConditionNet
    from point_diffusion_refinement_amd.pointnet2.models.pointnet2_with_pcld_condition import PointNet2CloudCondition
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = PointNet2CloudCondition(ddpm_pointnet_config()).eval().to(dev)
    fused = FusedCloudConditionNet(net, precision=os.environ.get("VARIANT_PRECISION", "f32"))
    x, cond, label = synthetic_batch(2, seed=3, device=dev)
    ts = torch.tensor([500.0, 20.0], device=dev)
    with torch.no_grad():
        first = fused(x, cond, ts=ts, label=label, use_retained_condition_feature=True)
        cached = fused(x * 0.9, cond, ts=ts - 1, label=label, use_retained_condition_feature=True)
        # a MIXED neighbourhood plan (x_t = q_sample(torus, 75): 30-90 % of the tiles walked): the variants of the
        # deduplicated evaluation differ only where walked and skipped tiles sit side by side
        from point_diffusion_refinement_amd.pointnet2 import util
        from point_diffusion_refinement_amd.pointnet2.configs import DIFFUSION_CONFIG, q_sample, synthetic_surface_batch
        x0, cond2, label2 = synthetic_surface_batch(2, seed=3, device=dev)
        xm = q_sample(x0, 75, util.calc_diffusion_hyperparams(**DIFFUSION_CONFIG), seed=3)
        fused.reset_cond_features()
        t75 = torch.full((2,), 75.0, device=dev)
        fused(xm, cond2, ts=t75, label=label2, use_retained_condition_feature=True)
        mixed = fused(xm * 0.98, cond2, ts=t75 - 1, label=label2, use_retained_condition_feature=True)
    torch.cuda.synchronize()
    torch.save({"first": first.cpu(), "cached": cached.cpu(), "mixed": mixed.cpu()}, sys.argv[1])


if __name__ == "__main__":
    main()
