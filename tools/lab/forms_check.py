"""Samples of the three step forms (adaptive / once / whole) after a 6-step restart on a surface, as
test_adaptive_sampler_picks_the_form_of_each_step_from_the_probe runs them: pairwise relative differences and how many
coordinates carry them.   python -m tools.lab.forms_check out.pt [other.pt]"""
import sys

import torch

from point_diffusion_refinement_amd.pointnet2 import fused_network as FN
from point_diffusion_refinement_amd.pointnet2 import util
from point_diffusion_refinement_amd.pointnet2.configs import DIFFUSION_CONFIG, ddpm_pointnet_config, synthetic_surface_batch
from point_diffusion_refinement_amd.pointnet2.models.pointnet2_with_pcld_condition import PointNet2CloudCondition
from point_diffusion_refinement_amd.pointnet2.reverse_sampler import GraphedReverseSampler


def rel(a, b):
    return ((a - b).abs() / (b.abs() + 1.0))


def main():
    cuda = torch.device("cuda:0")
    torch.manual_seed(0)
    net = PointNet2CloudCondition(ddpm_pointnet_config()).eval().to(cuda)
    fused = FN.FusedCloudConditionNet(net)
    x0, cond, label = synthetic_surface_batch(2, seed=3, device=cuda)
    dh = util.calc_diffusion_hyperparams(**DIFFUSION_CONFIG)
    outs = {}
    for steps in (1, 2, 6):
        for mode in ("adaptive", "once", "whole"):
            s = GraphedReverseSampler(fused, dh, noise='cpu', use_graph=True, neighbourhoods=mode)
            torch.manual_seed(7)
            outs[(mode, steps)] = s.sample((2, 2048, 3), cond, label, use_a_precomputed_XT=True, step=steps, XT=x0).clone()
        for a, b in (("adaptive", "once"), ("adaptive", "whole"), ("once", "whole")):
            e = rel(outs[(a, steps)], outs[(b, steps)])
            print("steps %d  %-8s vs %-6s max %.3e  mean %.3e  coords > 1e-4: %d  > 1e-5: %d of %d" % (
                steps, a, b, float(e.max()), float(e.mean()), int((e > 1e-4).sum()), int((e > 1e-5).sum()), e.numel()))
    torch.save({k: v.cpu() for k, v in outs.items()}, sys.argv[1])
    if len(sys.argv) > 2:
        other = torch.load(sys.argv[2])
        for k in sorted(outs):
            e = rel(outs[k].cpu(), other[k])
            print("vs %s  %s: max %.3e mean %.3e  > 1e-4: %d" % (sys.argv[2], k, float(e.max()), float(e.mean()),
                                                                 int((e > 1e-4).sum())))


if __name__ == "__main__":
    main()
