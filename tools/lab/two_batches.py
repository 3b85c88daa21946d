"""Experiment: ONE B = 32 sampler vs TWO independent B = 32 samplers in flight on one GPU (own
network copy, own captured graphs, own streams).  Round 3 measured two B = 16 HALVES (11.07 vs 8.71 ms per 32-cloud step:
no overlap, every kernel filled the chip); since round 4 the step is a chain of small launches for most of its length
(profiles/r5_timeline.json: 59 % of the span with one kernel in flight), so two whole batches may interleave.
    python -m tools.lab.two_batches [steps]
Prints cloud-steps/s of the job (all clouds of all samplers / wall time)."""
import json
import sys
import time

import torch

import bench as BN
from point_diffusion_refinement_amd.pointnet2.configs import synthetic_batch


def run(nsamplers, steps, neighbourhoods="adaptive"):
    dev = torch.device("cuda:0")
    B = BN.B_PER_GPU
    samplers, streams = [], []
    for i in range(nsamplers):
        s, _ = BN.build_sampler(dev, True, fused=True, neighbourhoods=neighbourhoods)
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
    return {"samplers": nsamplers, "batch_each": B, "steps_each": steps, "form": neighbourhoods,
            "ms_per_round": dt / steps * 1e3, "cloud_steps_per_s": nsamplers * B * steps / dt}


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    out = []
    for form in ("adaptive", "whole"):
        for n in (1, 2, 3, 1, 2):
            r = run(n, steps, form)
            out.append(r)
            print(json.dumps(r), flush=True)
