"""Bit-identity probe for scheduling knobs of the layer kernels (tile order, grid size): runs 1 uncached + 3 cached
eager reverse steps of the DDPM config at B = 16 with CPU-seeded inputs and writes x to the given file; two runs under
different knob settings must produce IDENTICAL bytes (the knobs change which workgroup computes a tile, not the tile).
    PDR_WS_XCD_ORDER=2 python -m tools.lab.order_check /tmp/a.pt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import build_sampler  # noqa: E402
from point_diffusion_refinement_amd.pointnet2.configs import synthetic_batch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    sampler, _ = build_sampler(dev, use_graph=False)
    sampler.noise = 'cpu'
    B = int(os.environ.get("ORDER_CHECK_B", "16"))
    x_T, cond, label = synthetic_batch(B, seed=0, device=dev)
    torch.manual_seed(1)
    sampler.begin((B, 2048, 3), cond, label, x_T=x_T)
    sampler.advance(3)
    torch.cuda.synchronize()
    torch.save(sampler._x.cpu(), sys.argv[1])
    if len(sys.argv) > 2:
        ref = torch.load(sys.argv[2])
        same = torch.equal(ref, sampler._x.cpu())
        print("identical to %s: %s (max |diff| %.3e)" % (sys.argv[2], same, float((ref - sampler._x.cpu()).abs().max())))


if __name__ == "__main__":
    main()
