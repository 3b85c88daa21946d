"""Experiment: one B=32 batch vs two concurrent B=16 half-batches (own network copy, own captured graph, own
stream).  Same total work; the question is how much of the step is latency / tail that a second stream hides.
    python -m tools.lab.two_stream"""
import time

import torch

import bench as BN
from point_diffusion_refinement_amd.pointnet2.configs import synthetic_batch


def run(nsplit, steps=20):
    dev = torch.device("cuda:0")
    B = 32 // nsplit
    samplers, streams = [], []
    for i in range(nsplit):
        s, _ = BN.build_sampler(dev, True, fused=True)
        x_T, cond, label = synthetic_batch(B, BN.N_POINTS, BN.M_COND, seed=i, device=dev)
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            s.begin((B, BN.N_POINTS, 3), cond, label, x_T=x_T)
            s.begin((B, BN.N_POINTS, 3), cond, label, x_T=x_T)
            s.advance(3)
        samplers.append(s)
        streams.append(st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for s, st in zip(samplers, streams):
            with torch.cuda.stream(st):
                s.advance(1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d x B=%d: %.3f ms per (32-cloud) step, %.1f cloud-steps/s" % (nsplit, B, dt / steps * 1e3,
                                                                       32 * steps / dt))


if __name__ == "__main__":
    run(1)
    run(2)
    run(4)
