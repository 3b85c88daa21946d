"""The once-per-batch FIRST reverse step (condition branch + global PointNet + embedding rows; VERDICT r3 missing 5) by
itself: `python -m tools.first_step_profile [B]` loads B synthetic clouds and calls sampler.begin() five times
(warm-up + 4 timed); under `rocprofv3 --kernel-trace --stats` the kernel table shows what a first step launches --
no Cijk_* (hipBLASLt), miopen* or naive_conv* rows since round 4."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_sampler  # noqa: E402
from point_diffusion_refinement_amd.pointnet2.configs import synthetic_batch  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda:0")
    sampler, _ = build_sampler(dev, use_graph=False)
    x_T, cond, label = synthetic_batch(B, seed=0, device=dev)
    sampler.begin((B, 2048, 3), cond, label, x_T=x_T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        sampler.begin((B, 2048, 3), cond, label, x_T=x_T)
    torch.cuda.synchronize()
    print("first (uncached) step, B = %d: %.2f ms" % (B, (time.perf_counter() - t0) / 4 * 1e3))


if __name__ == "__main__":
    main()
