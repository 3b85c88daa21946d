#!/usr/bin/env python
"""bench.py -- DDPM reverse steps/s of PDR's dual-path PointNet++ on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed environment: bench.py re-launches itself under
`python -m torch.distributed.run --nproc-per-node N` (one process per GPU, rendezvous on
127.0.0.1); launched by the driver under torch.distributed.run it reads RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment.

A "step" is ONE reverse-diffusion step (cached-condition eps-network forward + DDPM update +
noise) over ONE batch of B=32 synthetic clouds (x_t (32,2048,3), condition (32,3072,4)), the
BASELINE.json configs[1] workload, fp32, random-init weights of the shipped DDPM architecture.
Each rank owns its own batch (weak scaling, no data-path collective); the K timed steps are
bracketed by barrier + synchronize and the MAX over ranks is reported.  After the timed region
every rank evaluates Chamfer / F1 / EMD of its batch against synthetic ground truth and the
(n, 5) metric records travel through ONE all-gather (generation.gather_records: RCCL for N > 1),
exactly the collective the generation harness uses at the end of a job.
Rank 0 prints one JSON line.  At N=1 rank 0 also reports
  * `roofline`: the dominant hand-written kernel of THIS run (selected from HIP-event timings of
    every C-ABI launch of an eager step), against its roof;
  * `cpu_baseline`: the CPU port (this repo's PyTorch-CPU network over the C oracle ops) on
    the host cores, on a bounded sample of the same workload;
  * `split_f16`: the same step with the wide GEMMs in the opt-in f16x3 MFMA mode (labelled,
    never the headline);
  * `config3_eval` / `config5_fastdpm_refine`: BASELINE configs[2] and configs[4] on one GPU with
    their own bounded CPU baselines.
"""
import argparse
import json
import os
import socket
import subprocess
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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="clouds per GPU (BASELINE: 32)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--unfused", action="store_true",
                    help="layer-by-layer PyTorch execution over the native ops instead of the fused kernels")
    ap.add_argument("--precision", choices=("f32", "split_f16", "split_bf16"), default="f32",
                    help="arithmetic of the wide 1x1-conv GEMMs of the HEADLINE run (default exact fp32 MFMA); "
                         "split_bf16 = deprecated alias of split_f16 (the halves are f16 since round 3)")
    ap.add_argument("--neighbourhoods", choices=("adaptive", "once", "whole"), default="adaptive",
                    help="form of the captured step of the HEADLINE run: adaptive (the product default: per-step switch), "
                         "once / whole = one form for every step (A/B and profiling)")
    ap.add_argument("--single-stream", action="store_true",
                    help="profiling: the two halves of every block on ONE stream, so that every kernel has the chip to "
                         "itself (tools/profile_round.sh's serial pass); never the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the live roofline of the dominant kernel")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the split-f16 / config-3 / config-5 legs (N=1 only)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline sample")
    ap.add_argument("--dry", action="store_true",
                    help="host-logic check without a GPU (tests): gloo backend, no sampler, synthetic metric "
                         "records through the same gather")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_spawn(args):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks ourselves."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def build_sampler(device, use_graph, fused=True, precision="f32", single_stream=False, neighbourhoods="adaptive"):
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
        model = FusedCloudConditionNet(net, precision=precision)
        model.two_streams = not single_stream
    return GraphedReverseSampler(model, dh, noise='device', use_graph=use_graph, neighbourhoods=neighbourhoods), net


def cpu_baseline(budget_s):
    """CPU port = BASELINE configs[0]: B = 1, N = 2048, 3072-point condition, the T = 50 p_sample loop
    (`util.sampling` over a T = 50 schedule: 1 uncached + 49 cached network calls + the reverse updates) with this
    repo's PyTorch-CPU fp32 network over the scalar C oracle ops (oracle/pdr_oracle.c).
    The intra-op thread count is TUNED first: on a 256-thread host the default (128 threads) oversubscribes these small
    GEMMs -- round 3 reported 0.62 cloud-steps/s there, 2.7x slower than the reference's Python measured on 8 cores
    (1.66) -- so one cached call is timed at 8 / 16 / 32 / 64 / default threads and the loop runs at the
    best count (`cores`).  The loop is run in full when it fits 3x the budget, otherwise cut and extrapolated (said so
    in `sample`)."""
    from point_diffusion_refinement_amd.pointnet2 import util
    from point_diffusion_refinement_amd.pointnet2.configs import ddpm_pointnet_config, synthetic_batch
    from point_diffusion_refinement_amd.pointnet2.models.pointnet2_with_pcld_condition import \
        PointNet2CloudCondition
    from tests.oracle_backend import oracle_ops
    torch.manual_seed(0)
    net = PointNet2CloudCondition(ddpm_pointnet_config()).eval()
    T = 50
    dh = util.calc_diffusion_hyperparams(T, 1e-4, 0.02)
    x0, cond, label = synthetic_batch(1, N_POINTS, M_COND, seed=0)
    default_threads = torch.get_num_threads()
    sweep = {}
    with torch.no_grad(), oracle_ops():
        # ---- thread sweep: one uncached call to fill the cache, then one timed cached call per candidate
        net(x0, cond, ts=torch.full((1,), float(T - 1)), label=label, use_retained_condition_feature=True)
        for nt in sorted({n for n in (8, 16, 32, 64, default_threads) if n <= max(default_threads, 8)}):
            torch.set_num_threads(nt)
            net(x0, cond, ts=torch.full((1,), float(T - 2)), label=label, use_retained_condition_feature=True)
            t0 = time.perf_counter()
            net(x0, cond, ts=torch.full((1,), float(T - 2)), label=label, use_retained_condition_feature=True)
            sweep[nt] = time.perf_counter() - t0
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        net.reset_cond_features()
        # ---- the T = 50 loop (util.py:184-255 order: eps, update, noise for t > 0)
        x = x0.clone()
        t0 = time.perf_counter()
        first = cached_s = 0.0
        n_cached = 0
        for t in range(T - 1, -1, -1):
            t1 = time.perf_counter()
            eps = net(x, cond, ts=torch.full((1,), float(t)), label=label, use_retained_condition_feature=True)
            x = (x - (1 - dh["Alpha"][t]) / torch.sqrt(1 - dh["Alpha_bar"][t]) * eps) / torch.sqrt(dh["Alpha"][t])
            if t > 0:
                x = x + dh["Sigma"][t] * torch.normal(0, 1, size=x.shape)
            dt = time.perf_counter() - t1
            if t == T - 1:
                first = dt
            else:
                cached_s += dt
                n_cached += 1
            if time.perf_counter() - t0 > 3.0 * budget_s and n_cached >= 3:
                break
        loop_s = time.perf_counter() - t0
        net.reset_cond_features()
    torch.set_num_threads(default_threads)
    cached = cached_s / n_cached
    full = n_cached == T - 1
    return {"value": round(1.0 / cached, 3), "unit": "cloud-steps/s", "cores": best, "kind": "port",
            "t50_loop_s": round(loop_s if full else first + (T - 1) * cached, 1), "t50_loop_measured": full,
            "thread_sweep_s_per_cached_call": {str(k): round(v, 3) for k, v in sweep.items()},
            "host_threads": os.cpu_count(),
            "sample": "BASELINE configs[0]: B=1, N=2048, 3072-pt condition, T=50 loop %s: 1 uncached call (%.2f s) + %d "
                      "cached calls (%.3f s each) at %d intra-op threads (best of the sweep), fp32 PyTorch-CPU network "
                      "over oracle/pdr_oracle.c ops" % ("run in full" if full else "cut at the budget, extrapolated",
                                                       first, n_cached, cached, best)}


def metric_records(x, label, seed):
    """Per-sample [cd_t, cd_p, f1, emd, label] of this rank's batch against synthetic ground truth, computed
    as the harness does (completion_eval.py:196-232: /2/scale, Chamfer + F1 at 1e-4, EMD)."""
    from point_diffusion_refinement_amd.pointnet2 import generation as G
    g = torch.Generator().manual_seed(10_000 + seed)
    gt = (torch.rand(x.shape[0], x.shape[1], 3, generator=g) * 2 - 1).to(x.device)
    _, rec = G.evaluate_batch(lambda c, l: x, None, label, gt)
    return rec


def config3_eval(device, cpu_seconds):
    """BASELINE configs[2]: Chamfer + F1 and approximate EMD on 10k synthetic (2048,3) pairs (1000-pair batches),
    plus the C oracle on a bounded sample of the same pairs."""
    import numpy as np
    from oracle import pdr_oracle as O   # cpu_baseline leg + checker only
    from point_diffusion_refinement_amd.pointnet2 import emd
    from point_diffusion_refinement_amd.pointnet2.chamfer_loss_new import calc_cd
    g = torch.Generator().manual_seed(0)
    nb, n, pairs = 1000, N_POINTS, 10000
    a = (torch.rand(nb, n, 3, generator=g) - 0.5).to(device)
    b = (torch.rand(nb, n, 3, generator=g) - 0.5).to(device)
    out = {"pairs": pairs, "points": n}
    for name, fn in (("chamfer_f1", lambda: calc_cd(a, b, calc_f1=True)),
                     ("emd", lambda: emd.earth_mover_distance(a, b))):
        for _ in range(3):                                # warm-up: allocator, clocks
            fn()
        torch.cuda.synchronize(device)
        # whole passes over the 10k pairs until at least 0.5 s have been timed (one Chamfer pass is ~14 ms: a single
        # pass was too short a window -- the driver saw 362 k pairs/s where longer runs give twice that)
        passes, t0 = 0, time.perf_counter()
        while passes == 0 or time.perf_counter() - t0 < 0.5:
            for _ in range(pairs // nb):
                fn()
            torch.cuda.synchronize(device)
            passes += 1
        out[name + "_pairs_per_s"] = round(passes * pairs / (time.perf_counter() - t0), 1)
        out[name + "_timed_passes"] = passes
    # the 1000-pair EMD batches against the C oracle on 8 pairs spread over a batch (first / middle / last: the kernels
    # stride pairs over the grid, emd_kernel.cu:41,174-196 `for i = blockIdx.x; i < b; i += gridDim.x`), north_star's 1e-4
    probe_pairs = [0, 1, nb // 3, nb // 2, nb // 2 + 1, 2 * nb // 3, nb - 2, nb - 1]
    emd_gpu = emd.earth_mover_distance(a, b).cpu().numpy()
    an_, bn_ = a.cpu().numpy(), b.cpu().numpy()
    from concurrent.futures import ThreadPoolExecutor as _Pool
    with _Pool(len(probe_pairs)) as _ex:                    # (~9 s per pair on one core: all eight at once)
        emd_ref = np.array(list(_ex.map(lambda i: float(O.emd(an_[i:i + 1], bn_[i:i + 1])[0]), probe_pairs)))
    np.testing.assert_allclose(emd_gpu[probe_pairs], emd_ref, rtol=1e-4)
    out["emd_checked_pairs"] = {"pairs_of_a_1000_pair_batch": probe_pairs, "rtol": 1e-4,
                                "max_rel": float(np.abs(emd_gpu[probe_pairs] / emd_ref - 1).max())}
    out["chamfer_valu_tflops"] = round(out["chamfer_f1_pairs_per_s"] * 2 * n * n * 8 / 1e12, 2)
    out["emd_texp_per_s"] = round(out["emd_pairs_per_s"] * 30 * n * n / 1e12, 3)
    # CPU: the scalar C oracle, one pair per call, calls spread over a thread pool of one worker per PHYSICAL core
    # (ctypes releases the GIL inside the C function).  Whole rounds of 2 pairs per worker are timed until at least
    # 3 s (Chamfer) / half the CPU budget (EMD) have passed: round 3 timed ONE round of 100 pairs on 256 workers,
    # i.e. the latency of the slowest call, and the builder's and the driver's runs differed 2x.
    from concurrent.futures import ThreadPoolExecutor
    cores = max(1, (os.cpu_count() or 2) // 2)
    n_cd = 100
    an, bn = a[:n_cd].cpu().numpy(), b[:n_cd].cpu().numpy()
    with ThreadPoolExecutor(cores) as ex:
        res = list(ex.map(lambda i: O.chamfer(bn[i:i + 1], an[i:i + 1]), range(n_cd)))      # checked + warm-up
        cd_ref = np.array([r[0].mean() + r[2].mean() for r in res])
        cd_t = calc_cd(a[:n_cd], b[:n_cd])[1].cpu().numpy()
        np.testing.assert_allclose(cd_t, cd_ref, rtol=1e-5)
        k_cd, t0 = 0, time.time()
        while time.time() - t0 < 3.0:
            list(ex.map(lambda i: O.chamfer(bn[i % n_cd:i % n_cd + 1], an[i % n_cd:i % n_cd + 1]),
                        range(k_cd, k_cd + 2 * cores)))
            k_cd += 2 * cores
        t_cd = (time.time() - t0) / k_cd
        k, t0 = 0, time.time()
        while k == 0 or time.time() - t0 < cpu_seconds / 2:
            list(ex.map(lambda i: O.emd(an[i % n_cd:i % n_cd + 1], bn[i % n_cd:i % n_cd + 1]), range(k, k + cores)))
            k += cores
        t_emd = (time.time() - t0) / k
    out["cpu_baseline"] = {"chamfer_pairs_per_s": round(1.0 / t_cd, 2), "emd_pairs_per_s": round(1.0 / t_emd, 3),
                           "cores": cores, "kind": "port",
                           "sample": "oracle/pdr_oracle.c (scalar C, one pair per call on a %d-worker pool = one per "
                                     "physical core): Chamfer %d pair evaluations over >= 3 s, EMD %d over >= %.0f s "
                                     "(pairs drawn from the first %d of the 10k)" % (cores, k_cd, k, cpu_seconds / 2,
                                                                                    n_cd)}
    return out


def config5_fastdpm_refine(device, B):
    """BASELINE configs[4] on one GPU: FastDPM S=50 (VAR, quadratic, kappa 0.5, hipGraph) coarse generation, ONE
    refinement forward with x8 point upsampling to N=16384, Chamfer at 16384 points; random-init networks."""
    from point_diffusion_refinement_amd.pointnet2 import generation as G
    from point_diffusion_refinement_amd.pointnet2 import util
    from point_diffusion_refinement_amd.pointnet2.chamfer_loss_new import calc_cd
    from point_diffusion_refinement_amd.pointnet2.configs import (DIFFUSION_CONFIG, ddpm_pointnet_config,
                                                                  refinement_pointnet_config, synthetic_batch)
    from point_diffusion_refinement_amd.pointnet2.fused_network import FusedCloudConditionNet
    from point_diffusion_refinement_amd.pointnet2.models.pointnet2_with_pcld_condition import \
        PointNet2CloudCondition
    from point_diffusion_refinement_amd.pointnet2.reverse_sampler import GraphedFastSampler
    torch.manual_seed(0)
    coarse_net = FusedCloudConditionNet(PointNet2CloudCondition(ddpm_pointnet_config()).eval().to(device))
    refine_net = FusedCloudConditionNet(PointNet2CloudCondition(refinement_pointnet_config(8)).eval().to(device))
    dh = util.calc_diffusion_hyperparams(**DIFFUSION_CONFIG)
    S = 50
    sampler = GraphedFastSampler(coarse_net, dh, DIFFUSION_CONFIG, length=S, sampling_method='var',
                                 schedule='quadratic', kappa=0.5, noise='device', use_graph=True)
    refiner = G.GraphedRefiner(refine_net, 0.001, 8)      # the refinement forward + x8 upsampling as one graph replay
    _, cond, label = synthetic_batch(B, seed=0, device=device)
    gt = torch.rand(B, 16384, 3, device=device) - 0.5
    out, reps = {}, 2
    with torch.no_grad():
        for rep in range(reps + 1):                       # first repetition = warm-up (capture, allocator)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            coarse = sampler.sample((B, N_POINTS, 3), cond, label)
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            fine = refiner(coarse, cond, label)
            torch.cuda.synchronize(device)
            t2 = time.perf_counter()
            cd_p, cd_t = calc_cd(fine / 2, gt)
            torch.cuda.synchronize(device)
            t3 = time.perf_counter()
            if rep:
                for k, v in (("coarse_s", t1 - t0), ("refine_s", t2 - t1), ("chamfer_16384_s", t3 - t2)):
                    out[k] = out.get(k, 0.0) + v / reps
    assert fine.shape == (B, 16384, 3) and bool(torch.isfinite(cd_t).all())
    total = sum(out.values())
    out = {k: round(v, 4) for k, v in out.items()}
    out.update(batch=B, fastdpm_steps=S, upsample=8, total_s=round(total, 4),
               completed_clouds_per_s=round(B / total, 2), refined_points_per_s=round(B * 16384 / total, 1),
               cpu_baseline="see cpu_baseline.t50_loop_s (B=1, 50 network calls on the host cores)")
    return out


def rank_times(elapsed, steps, world, device=None):
    """(max over ranks of `elapsed`, per-rank report) -- the contract's MAX, plus every rank's own ms per step so that the
    first real N-GPU run shows whether one straggler or the host side limits weak scaling."""
    if world == 1:
        return elapsed, {"ms_per_step": [round(elapsed / steps * 1e3, 4)], "min": round(elapsed / steps * 1e3, 4),
                         "max": round(elapsed / steps * 1e3, 4), "rank_of_max": 0}
    mine = torch.tensor([elapsed], dtype=torch.float64, **({"device": device} if device is not None else {}))
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    ms = [float(t.item()) / steps * 1e3 for t in every]
    tmax = mine.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    worst = max(range(world), key=lambda r: ms[r])
    return float(tmax.item()), {"ms_per_step": [round(m, 4) for m in ms], "min": round(min(ms), 4),
                                "max": round(max(ms), 4), "rank_of_max": worst}


def dry_main(args, world, rank):
    """Host logic of the N-rank bench without GPUs (CPU test): rendezvous, barrier, max-over-ranks timing and
    the metric-record all-gather run for real over gloo; the sampler is replaced by a sleep."""
    from point_diffusion_refinement_amd.pointnet2 import generation as G
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    B = args.batch
    t0 = time.perf_counter()
    time.sleep(0.01 * args.steps)
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    elapsed, per_rank = rank_times(elapsed, args.steps, world)
    n_local = B if rank < world - 1 or world == 1 else max(B - 3, 0)       # a short last shard, as in a real job
    rec = torch.rand(n_local, 5, generator=torch.Generator().manual_seed(rank))
    rec[:, 4] = float(rank)
    allrec, counts = G.gather_records(rec, return_counts=True)
    if rank == 0:
        print(json.dumps({"metric": "DDPM reverse steps/sec (T=1000, N=2048)", "dry": True, "n_gpus": world,
                          "value": round(world * B * args.steps / elapsed, 2), "steps": args.steps,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 4), "per_rank": per_rank,
                          "ranks_seen": int(sum(1 for c in counts if c > 0)), "records_gathered": int(allrec.shape[0]),
                          "world_size_after_gather": dist.get_world_size() if world > 1 else 1,
                          "records_per_rank": [int(c) for c in counts],
                          # rank order of the concatenation (generate_samples_distributed.py:84-95): the rank column of
                          # the gathered records, run-length encoded
                          "record_rank_runs": [[r, int((allrec[:, 4] == r).sum())] for r in
                                               sorted(set(allrec[:, 4].tolist()))],
                          "rank_column_sorted": bool((allrec[1:, 4] >= allrec[:-1, 4]).all()),
                          "record_rank_column": sorted(set(allrec[:, 4].tolist()))}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.precision == "split_bf16":
        print("bench.py: --precision split_bf16 is a deprecated alias of split_f16", file=sys.stderr)
        args.precision = "split_f16"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d (launch with --nproc-per-node %d, or without "
                         "torch.distributed.run and let bench.py start the ranks)" % (world, args.gpus, args.gpus))
    if args.dry:
        return dry_main(args, world, rank)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback of the product path)"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)

    from point_diffusion_refinement_amd.pointnet2 import generation as G
    from point_diffusion_refinement_amd.pointnet2.configs import synthetic_batch
    sampler, net = build_sampler(device, not args.no_graph, fused=not args.unfused, precision=args.precision,
                                 single_stream=args.single_stream, neighbourhoods=args.neighbourhoods)
    B = args.batch
    # every rank draws ITS OWN shard of the synthetic partial clouds (seed offset by rank)
    x_T, cond, label = synthetic_batch(B, N_POINTS, M_COND, seed=rank, device=device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    def timed_steps(smp, steps, warmup):
        smp.begin((B, N_POINTS, 3), cond, label, x_T=x_T)        # lazy init (allocator, library handles) untimed
        torch.cuda.synchronize(device)
        t0 = time.time()
        smp.begin((B, N_POINTS, 3), cond, label, x_T=x_T)        # first (uncached) step of a batch, eager
        torch.cuda.synchronize(device)
        first = time.time() - t0
        smp.advance(max(warmup, 1))                               # includes graph capture
        barrier()
        t0 = time.perf_counter()
        left = steps
        while left > 0:
            if smp.remaining == 0:
                # more steps asked for than one batch has (T = 1000): the job goes on with its next batch, whose
                # first, uncached step (begin) is one of the timed steps -- as in a real generation job
                smp.begin((B, N_POINTS, 3), cond, label, x_T=x_T)
                left -= 1
                continue
            n = min(left, smp.remaining)
            smp.advance(n)
            left -= n
        barrier()
        return time.perf_counter() - t0, first

    elapsed, first_step_s = timed_steps(sampler, args.steps, args.warmup)
    elapsed, per_rank = rank_times(elapsed, args.steps, world, device)
    x_now = sampler._x
    assert bool(torch.isfinite(x_now).all()), "non-finite samples"

    # ---- end-of-job metric reduction (outside the timed region): the ONE collective of the path
    t0 = time.perf_counter()
    rec = metric_records(x_now.clone(), label, rank)
    allrec, counts = G.gather_records(rec, return_counts=True)
    torch.cuda.synchronize(device)
    gather_s = time.perf_counter() - t0
    summary = G.summarize(allrec)

    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed                       # cloud-steps/s, whole job
    out = {
        "metric": "DDPM reverse steps/sec (T=1000, N=2048)", "value": round(value, 2), "unit": "cloud-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == "f32" else "f16x3 (wide GEMMs) + f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: B=%d/GPU, N=2048, 3072-pt mirrored condition, T=1000 DDPM "
                               "reverse sampling, random-init dual-path PointNet++ (9.76 M params), cached "
                               "condition step" % B,
                   "input": "x_T ~ N(0,1) clouds, U[-1,1]^3 mirrored condition (configs.synthetic_batch): the noise "
                            "regime of a reverse process; trajectory_weighted = the T=1000-weighted mean on a "
                            "trajectory that ends on a surface",
                   "global_batch": world * B, "parallelism": "dp%d" % world,
                   "launch": "eager" if args.no_graph else "hipGraph replay", "noise": "device Philox",
                   **({"single_stream": True} if args.single_stream else {}),
                   "execution": "layer-by-layer torch + native ops" if args.unfused else "fused channel-last HIP",
                   **({} if args.unfused else
                      {"neighbourhoods": "K identical rows of a <= 1-point ball evaluated once where the sampler's "
                                         "per-step probe says it pays (results unchanged; see step_form, "
                                         "one_point_neighbourhoods, trajectory)"})},
        "completed_points_per_s_per_gpu": round(value / world * N_POINTS / T_STEPS, 2),
        # every rank's own time over the same barrier-bracketed region (value uses the MAX)
        "per_rank": per_rank,
        "first_uncached_step_ms": round(first_step_s * 1e3, 2),
        # the REFERENCE's GEMM work per cloud-step (26.35 GFLOP) / time: an equivalent rate, NOT an achieved
        # one -- the executed flops are fewer (`executed_work` below)
        "reference_equivalent_tflops": round(value * GEMM_GFLOP_PER_CLOUD_STEP / 1e3, 2),
        "step_form": {"sampler": getattr(sampler, "neighbourhoods", None),
                      "replayed": dict(getattr(sampler, "mode_counts", {})),
                      "tiles_walked_frac": None if getattr(sampler, "walked_share", None) is None
                      else round(sampler.walked_share, 4)},
        "ranks_seen": int(sum(1 for c in counts if c > 0)), "records_gathered": int(allrec.shape[0]),
        # (after the gather: the collective's own view of the job -- the first real N-GPU run verifies itself)
        "world_size_after_gather": dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1,
        "metric_gather": {"collective": "all_gather of (n,5) f32 records [cd_t, cd_p, f1, emd, label]",
                          "backend": "nccl (RCCL)" if world > 1 else "none (1 rank)",
                          "eval_plus_gather_ms": round(gather_s * 1e3, 2),
                          "avg_cd": summary["avg_cd"], "avg_emd": summary["avg_emd"]},
    }
    solo = world == 1 and rank == 0
    if solo and not args.no_roofline:
        try:
            from tools.kernel_roofline import dominant_kernel_roofline
            if args.unfused:
                raise RuntimeError("roofline is reported for the fused execution only")
            out["roofline"] = dominant_kernel_roofline(sampler)
        except Exception as e:  # never lose the headline number to an instrumentation problem
            out["roofline"] = {"error": repr(e)}
    if solo and not args.no_extras and not args.unfused:
        # The headline evaluates neighbourhoods that are K copies of one row ONCE (fused_network.DEDUP) where
        # that pays: x_T and the x_t of this benchmark are noise, as they are for most of a reverse process, so most balls
        # hold at most one point and the sampler's per-step switch (reverse_sampler.py) replays the deduplicated step.
        # Reported next to it: the share of the per-neighbour tile work the step walked, the same step with every
        # neighbourhood evaluated (the cost of ANY input under the switch, + its probe), the executed GEMM work of both
        # forms, and -- `trajectory` -- both forms along a trajectory that ends on a surface.
        try:
            from tools.kernel_roofline import executed_gflop_per_step
            s4, _ = build_sampler(device, not args.no_graph, precision=args.precision, neighbourhoods="whole")
            el4, _ = timed_steps(s4, args.steps, args.warmup)
            ex = {}
            # the headline sampler is adaptive: its time is the 'once' form's only if every timed step replayed that
            # form -- otherwise the deduplicated form is timed on a sampler of its own
            ms_once = ms_per_step
            if sampler.mode_counts.get("whole", 0) > 0 or getattr(sampler, "neighbourhoods", "once") == "whole":
                s5, _ = build_sampler(device, not args.no_graph, precision=args.precision, neighbourhoods="once")
                el5, _ = timed_steps(s5, args.steps, args.warmup)
                ms_once = el5 / args.steps * 1e3
                del s5
            for mode in ("once", "whole"):
                g = executed_gflop_per_step(s4, mode)
                ms = ms_once if mode == "once" else el4 / args.steps * 1e3
                ex[mode] = {"executed_gflop_per_step": round(g, 1), "ms_per_step": round(ms, 4),
                            "executed_tflops": round(g / ms, 2), "step_mfma_frac": round(g / ms / 157.3, 4)}
            del s4
            out["one_point_neighbourhoods"] = {
                "evaluated_once": sampler.mode_counts.get("once", 0) > 0,
                "tiles_walked_frac": out["step_form"]["tiles_walked_frac"],
                "whole_evaluation": {"value": round(B * args.steps / el4, 2), "unit": "cloud-steps/s",
                                     "ms_per_step": round(el4 / args.steps * 1e3, 4)},
                "note": "ball_query pads a neighbourhood with its first hit: a query with <= 1 point in its ball "
                        "contributes K identical rows to every per-neighbour tensor of its block.  128-row tiles made of "
                        "such queries only are skipped by the per-neighbour launches; a per-query chain supplies their "
                        "GroupNorm moments (x K) and pooled rows -- same results (tests/test_fused_gpu.py::"
                        "test_one_point_neighbourhoods_*, the reference goldens).  tiles_walked_frac: share of the "
                        "deduplicated blocks' tiles this input needed; whole_evaluation: the step with nothing skipped "
                        "(what the sampler replays when the walked share exceeds its threshold)"}
            out["executed_work"] = {
                "headline_step": ex["once"], "whole_evaluation": ex["whole"],
                "reference_gflop_per_step": round(B * GEMM_GFLOP_PER_CLOUD_STEP, 1),
                "note": "sum over the layer launches of one step of 2 rows_walked Cin Cout (tools/kernel_roofline."
                        "executed_gflop_per_step); step_mfma_frac = executed TFLOP/s / 157.3 (fp32 MFMA peak): the "
                        "whole-step fraction next to roofline.frac of the dominant kernel"}
        except Exception as e:
            out["one_point_neighbourhoods"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        try:
            from tools import trajectory
            from point_diffusion_refinement_amd.pointnet2 import util
            from point_diffusion_refinement_amd.pointnet2.configs import DIFFUSION_CONFIG

            def fast():
                from point_diffusion_refinement_amd.pointnet2.reverse_sampler import GraphedFastSampler
                smp, _ = build_sampler(device, True, precision=args.precision)
                dh = util.calc_diffusion_hyperparams(**DIFFUSION_CONFIG)
                return GraphedFastSampler(smp.net, dh, DIFFUSION_CONFIG, length=50, sampling_method='var',
                                          schedule='quadratic', kappa=0.5, noise='device', use_graph=True)

            def refine():
                from point_diffusion_refinement_amd.pointnet2.configs import refinement_pointnet_config
                from point_diffusion_refinement_amd.pointnet2.fused_network import FusedCloudConditionNet
                from point_diffusion_refinement_amd.pointnet2.models.pointnet2_with_pcld_condition import \
                    PointNet2CloudCondition
                torch.manual_seed(0)
                return FusedCloudConditionNet(PointNet2CloudCondition(refinement_pointnet_config(8)).eval().to(device),
                                              precision=args.precision)
            out["trajectory"] = trajectory.measure(
                lambda: build_sampler(device, True, precision=args.precision)[0], fast, refine, device, B,
                util.calc_diffusion_hyperparams(**DIFFUSION_CONFIG))
            # the per-step cost of a REAL sampling job (its x_t ends on a surface), next to `value` (measured on the
            # input BASELINE names: x_T ~ N(0,1) clouds, the noise regime most of a reverse process is in)
            wm = out["trajectory"]["ddpm_t1000"]["weighted_mean_ms"]["adaptive"]
            out["trajectory_weighted"] = {"ms_per_step": wm, "value": round(B / wm * 1e3, 1), "unit": "cloud-steps/s",
                                          "completed_points_per_s_per_gpu": round(B / wm * 1e3 * N_POINTS / T_STEPS, 1),
                                          "note": "T = 1000-weighted mean of the step time along x_t = q_sample(torus, t) "
                                                  "with the sampler's per-step switch (trajectory.ddpm_t1000)"}
        except Exception as e:
            out["trajectory"] = {"error": repr(e)}
        torch.cuda.empty_cache()
    if solo and not args.no_extras and not args.unfused and not args.no_graph:
        # job-level throughput with TWO independent B = 32 batches in flight on this GPU (own network copy, own graphs, own
        # streams): VERDICT r5 item 7.  Not the headline -- and, measured, not a gain: the step's four streams are the
        # device's four hardware queues, a second graph's launches queue behind the first's (tools/lab/two_batches.py).
        try:
            pair, streams = [], []
            for k in range(2):
                sk, _ = build_sampler(device, True, precision=args.precision, neighbourhoods=args.neighbourhoods)
                xk, ck, lk = synthetic_batch(B, N_POINTS, M_COND, seed=100 + k, device=device)
                stk = torch.cuda.Stream(device=device)
                with torch.cuda.stream(stk):
                    sk.begin((B, N_POINTS, 3), ck, lk, x_T=xk)
                    sk.advance(3)
                pair.append(sk)
                streams.append(stk)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                for sk, stk in zip(pair, streams):
                    with torch.cuda.stream(stk):
                        sk.advance(1)
            torch.cuda.synchronize(device)
            el = time.perf_counter() - t0
            out["two_batches_in_flight"] = {
                "value": round(2 * B * args.steps / el, 2), "unit": "cloud-steps/s",
                "ms_per_step_pair": round(el / args.steps * 1e3, 4), "vs_one_batch": round(2 * B * args.steps / el / value, 4),
                "note": "two B=32 samplers replaying their captured steps on two streams; the headline `value` is one "
                        "B=32 graph"}
            del pair, streams
        except Exception as e:
            out["two_batches_in_flight"] = {"error": repr(e)}
        torch.cuda.empty_cache()
    if solo and not args.no_extras and not args.unfused and args.precision == "f32":
        try:
            s2, _ = build_sampler(device, not args.no_graph, precision="split_f16")
            el2, _ = timed_steps(s2, args.steps, args.warmup)
            out["split_f16"] = {"value": round(B * args.steps / el2, 2), "unit": "cloud-steps/s",
                                 "ms_per_step": round(el2 / args.steps * 1e3, 4),
                                 "completed_points_per_s_per_gpu": round(B * args.steps / el2 * N_POINTS / T_STEPS, 2),
                                 "dtype": "f16x3 MFMA (both operands as f16 hi + lo, 3 products, fp32 accumulate: ~22 "
                                          "mantissa bits) for the 128- / 64-column GEMM tiles with Cin >= 64, exact fp32 "
                                          "MFMA elsewhere",
                                 "note": "opt-in precision='split_f16', NOT the headline; held to the exact mode's "
                                         "parity bars (tests/test_fused_gpu.py::test_split_f16_*, the full-size "
                                         "reference goldens of tests/test_reference_golden.py; margins in "
                                         "profiles/r3_parity.json)"}
            del s2
        except Exception as e:
            out["split_f16"] = {"error": repr(e)}
        torch.cuda.empty_cache()
        for key, fn in (("config3_eval", lambda: config3_eval(device, args.cpu_seconds)),
                        ("config5_fastdpm_refine", lambda: config5_fastdpm_refine(device, B))):
            try:
                out[key] = fn()
            except Exception as e:
                out[key] = {"error": repr(e)}
            torch.cuda.empty_cache()
    if solo and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
