"""BASELINE configs[4] on one GPU: FastDPM S=50 (VAR, quadratic, kappa 0.5) coarse generation + ONE refinement forward
(include_t False, x8 point upsampling) + Chamfer at 16384 points, B=32 synthetic clouds, random-init networks.
    python -m tools.config5_bench   -> one JSON line"""
import json
import time

import torch

from point_diffusion_refinement_amd.pointnet2 import generation as G
from point_diffusion_refinement_amd.pointnet2 import util
from point_diffusion_refinement_amd.pointnet2.chamfer_loss_new import calc_cd
from point_diffusion_refinement_amd.pointnet2.configs import (DIFFUSION_CONFIG, ddpm_pointnet_config,
                                                              refinement_pointnet_config, synthetic_batch)
from point_diffusion_refinement_amd.pointnet2.fused_network import FusedCloudConditionNet
from point_diffusion_refinement_amd.pointnet2.models.pointnet2_with_pcld_condition import PointNet2CloudCondition
from point_diffusion_refinement_amd.pointnet2.reverse_sampler import GraphedFastSampler


def main(B=32, S=50, reps=3):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    coarse_net = FusedCloudConditionNet(PointNet2CloudCondition(ddpm_pointnet_config()).eval().to(dev))
    refine_net = FusedCloudConditionNet(PointNet2CloudCondition(refinement_pointnet_config(8)).eval().to(dev))
    dh = util.calc_diffusion_hyperparams(**DIFFUSION_CONFIG)
    sampler = GraphedFastSampler(coarse_net, dh, DIFFUSION_CONFIG, length=S, sampling_method='var',
                                 schedule='quadratic', kappa=0.5, noise='device', use_graph=True)
    _, cond, label = synthetic_batch(B, seed=0, device=dev)
    gt = torch.rand(B, 16384, 3, device=dev) - 0.5
    out = {}
    with torch.no_grad():
        for rep in range(reps + 1):                       # first repetition = warm-up (capture, MIOpen, allocator)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            coarse = sampler.sample((B, 2048, 3), cond, label)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            fine = G.refine_completion(refine_net, coarse, cond, label, 0.001, 8)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            cd_p, cd_t = calc_cd(fine / 2, gt)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            if rep:
                for k, v in (("coarse_s", t1 - t0), ("refine_s", t2 - t1), ("chamfer_16384_s", t3 - t2)):
                    out[k] = out.get(k, 0.0) + v / reps
    assert fine.shape == (B, 16384, 3) and bool(torch.isfinite(cd_t).all())
    total = sum(out.values())
    out = {k: round(v, 4) for k, v in out.items()}
    out.update(batch=B, fastdpm_steps=S, upsample=8, total_s=round(total, 4),
               completed_clouds_per_s=round(B / total, 2), completed_points_per_s=round(B * 16384 / total, 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
