"""What would a reverse step cost if the per-neighbourhood rows that only DUPLICATE a query's first neighbour were not
computed?  Upper-bound probe (NOT the product network): the x_t branch and the feature-transfer blocks are built with
nsample = 8 / 16 instead of 32 (the cached condition branch keeps 32), everything else as in bench.py.
    python -m tools.lab.nsample_upper_bound"""
import json
import time

import torch

import bench
from point_diffusion_refinement_amd.pointnet2 import configs as C
from point_diffusion_refinement_amd.pointnet2 import util
from point_diffusion_refinement_amd.pointnet2.fused_network import FusedCloudConditionNet
from point_diffusion_refinement_amd.pointnet2.models.pointnet2_with_pcld_condition import PointNet2CloudCondition
from point_diffusion_refinement_amd.pointnet2.reverse_sampler import GraphedReverseSampler


def main():
    dev = torch.device("cuda", 0)
    x_T, cond, label = C.synthetic_batch(32, bench.N_POINTS, bench.M_COND, seed=0, device=dev)
    out = {}
    for ns in (32, 16, 8):
        cfg = C.ddpm_pointnet_config()
        cfg["architecture"]["nsample"] = [ns] * 4
        cfg["feature_mapper_architecture"]["encoder_nsample"] = [ns] * 4
        cfg["feature_mapper_architecture"]["decoder_nsample"] = [ns] * 5
        torch.manual_seed(0)
        net = PointNet2CloudCondition(cfg).to(dev).eval()
        dh = util.calc_diffusion_hyperparams(**C.DIFFUSION_CONFIG)
        smp = GraphedReverseSampler(FusedCloudConditionNet(net), dh, noise='device', use_graph=True)
        smp.begin((32, bench.N_POINTS, 3), cond, label, x_T=x_T)
        smp.advance(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        smp.advance(40)
        torch.cuda.synchronize()
        out["nsample_%d" % ns] = round((time.perf_counter() - t0) / 40 * 1e3, 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
