#!/usr/bin/env python
"""bench.py -- DDPM reverse steps/s of PDR's dual-path PointNet++ on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE reverse-diffusion step (cached-condition eps-network forward + DDPM update +
noise) over ONE batch of B=32 synthetic clouds (x_t (32,2048,3), condition (32,3072,4)), the
BASELINE.json configs[1] workload, fp32, random-init weights of the shipped DDPM architecture.
Each rank owns its own batch (weak scaling, no data-path collective); the K timed steps are
bracketed by barrier + synchronize and the MAX over ranks is reported.  Rank 0 prints one JSON
line.  At N=1 rank 0 also reports
  * `roofline`: the dominant hand-written kernel, timed with HIP events on the launch stream;
  * `cpu_baseline`: the CPU port (this repo's PyTorch-CPU network over the C oracle ops) on
    the host cores, on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

B_PER_GPU = 32
N_POINTS, M_COND, T_STEPS = 2048, 3072, 1000
GEMM_GFLOP_PER_CLOUD_STEP = 26.35          # 154 1x1 convs, cached condition
HBM_PEAK_GBS, FP32_PEAK_TFLOPS = 8000.0, 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="clouds per GPU (BASELINE: 32)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--unfused", action="store_true",
                    help="layer-by-layer PyTorch execution over the native ops instead of the fused kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the live roofline of the dominant kernel")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline sample")
    return ap.parse_args()


def build_sampler(device, use_graph, fused=True):
    from point_diffusion_refinement_amd.pointnet2 import util
    from point_diffusion_refinement_amd.pointnet2.configs import DIFFUSION_CONFIG, ddpm_pointnet_config
    from point_diffusion_refinement_amd.pointnet2.models.pointnet2_with_pcld_condition import \
        PointNet2CloudCondition
    from point_diffusion_refinement_amd.pointnet2.reverse_sampler import GraphedReverseSampler
    torch.manual_seed(0)
    net = PointNet2CloudCondition(ddpm_pointnet_config()).to(device).eval()
    dh = util.calc_diffusion_hyperparams(**DIFFUSION_CONFIG)
    model = net
    if fused:
        from point_diffusion_refinement_amd.pointnet2.fused_network import FusedCloudConditionNet
        model = FusedCloudConditionNet(net)
    return GraphedReverseSampler(model, dh, noise='device', use_graph=use_graph), net


def cpu_baseline(budget_s):
    """CPU port: the same network in PyTorch-CPU fp32 over the C oracle ops (oracle/pdr_oracle.c), B=1,
    1 uncached + as many cached reverse steps as fit in the budget (>= 3)."""
    from point_diffusion_refinement_amd.pointnet2 import util
    from point_diffusion_refinement_amd.pointnet2.configs import (DIFFUSION_CONFIG, ddpm_pointnet_config,
                                                                  synthetic_batch)
    from point_diffusion_refinement_amd.pointnet2.models.pointnet2_with_pcld_condition import \
        PointNet2CloudCondition
    from tests.oracle_backend import oracle_ops
    torch.manual_seed(0)
    net = PointNet2CloudCondition(ddpm_pointnet_config()).eval()
    dh = util.calc_diffusion_hyperparams(**DIFFUSION_CONFIG)
    x, cond, label = synthetic_batch(1, N_POINTS, M_COND, seed=0)
    n = 0
    with torch.no_grad(), oracle_ops():
        t0 = time.time()
        t = T_STEPS - 1
        ts = torch.full((1,), float(t))
        eps = net(x, cond, ts=ts, label=label, use_retained_condition_feature=True)      # uncached first step
        first = time.time() - t0
        t1 = time.time()
        while n < 3 or (time.time() - t0 < budget_s and n < 200):
            x = (x - (1 - dh["Alpha"][t]) / torch.sqrt(1 - dh["Alpha_bar"][t]) * eps) / torch.sqrt(dh["Alpha"][t])
            x = x + dh["Sigma"][t] * torch.normal(0, 1, size=x.shape)
            t -= 1
            eps = net(x, cond, ts=torch.full((1,), float(t)), label=label, use_retained_condition_feature=True)
            n += 1
        cached = (time.time() - t1) / n
    return {"value": round(1.0 / cached, 3), "unit": "cloud-steps/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "B=1: 1 uncached step (%.2f s) + %d cached steps (%.3f s each), N=2048, 3072-pt condition, "
                      "fp32 PyTorch-CPU network over oracle/pdr_oracle.c ops" % (first, n, cached)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback of the product path)"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from point_diffusion_refinement_amd.pointnet2.configs import synthetic_batch
    sampler, net = build_sampler(device, not args.no_graph, fused=not args.unfused)
    B = args.batch
    # every rank draws ITS OWN shard of the synthetic partial clouds (seed offset by rank)
    x_T, cond, label = synthetic_batch(B, N_POINTS, M_COND, seed=rank, device=device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    sampler.begin((B, N_POINTS, 3), cond, label, x_T=x_T)       # lazy init (MIOpen find, allocator) untimed
    torch.cuda.synchronize(device)
    t0 = time.time()
    sampler.begin((B, N_POINTS, 3), cond, label, x_T=x_T)       # first (uncached) step of a batch, eager
    torch.cuda.synchronize(device)
    first_step_s = time.time() - t0
    sampler.advance(max(args.warmup, 1))                          # includes graph capture
    barrier()
    t0 = time.perf_counter()
    sampler.advance(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    x_now = sampler._x
    assert bool(torch.isfinite(x_now).all()), "non-finite samples"

    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed                       # cloud-steps/s, whole job
    out = {
        "metric": "DDPM reverse steps/sec (T=1000, N=2048)", "value": round(value, 2), "unit": "cloud-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: B=%d/GPU, N=2048, 3072-pt mirrored condition, T=1000 DDPM "
                               "reverse sampling, random-init dual-path PointNet++ (9.76 M params), cached "
                               "condition step" % B,
                   "global_batch": world * B, "parallelism": "dp%d" % world,
                   "launch": "eager" if args.no_graph else "hipGraph replay", "noise": "device Philox",
                   "execution": "layer-by-layer torch + native ops" if args.unfused else "fused channel-last HIP"},
        "completed_points_per_s_per_gpu": round(value / world * N_POINTS / T_STEPS, 2),
        "first_uncached_step_ms": round(first_step_s * 1e3, 2),
        "gemm_tflops": round(value * GEMM_GFLOP_PER_CLOUD_STEP / 1e3, 2),
    }
    if world == 1 and rank == 0 and not args.no_roofline:
        try:
            from tools.kernel_roofline import dominant_kernel_roofline
            if args.unfused:
                raise RuntimeError("roofline is reported for the fused execution only")
            out["roofline"] = dominant_kernel_roofline(sampler)
        except Exception as e:  # never lose the headline number to an instrumentation problem
            out["roofline"] = {"error": repr(e)}
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
