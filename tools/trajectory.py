"""Step time along a reverse trajectory that converges to a surface (used by bench.py's `trajectory` leg).

The headline step of bench.py runs on x_T ~ N(0, 1) -- the input BASELINE.json names, and what x_t looks like for most of
a reverse process -- but the deduplicated step's cost depends on how many balls hold more than one point
(fused_network.DEDUP), and a real job ends on a surface.  Without a checkpoint the trajectory of a
trained network is not available, its MARGINAL is: x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) eps (reference
util.py:280-282).  This module times ONE captured step of each form (`once`: one-point neighbourhoods evaluated once,
`whole`: every neighbourhood evaluated) at x_t = q_sample(torus, t) over a grid of t, reads the walked share the step
itself publishes, applies the sampler's own switch rule, and integrates piecewise-linearly over the schedule:
    ddpm_t1000        the T = 1000 loop of BASELINE configs[1] (weight 1 per integer t)
    fastdpm_s50       the S = 50 quadratic schedule of configs[4] (GraphedFastSampler; one weight per step)
    dense_input       t = 0: the worst case of the deduplicated form, with the product default beside both
    refinement        the refinement forward of configs[4] on the finished surface, both forms (eager)
"""
import statistics
import time

import torch

from point_diffusion_refinement_amd.pointnet2.configs import q_sample, synthetic_surface_batch

T_GRID = (999, 800, 500, 300, 200, 150, 100, 75, 50, 35, 20, 10, 0)


def _time_replays(sampler, mode, x, counter, ts_value, reps):
    """Median ms of `reps` replays of the captured `mode` step from state (x, counter); the walked share it published."""
    dev = sampler.device
    sampler._mode = mode
    if mode not in sampler._graphs:
        sampler._x.copy_(x)
        sampler._t.fill_(counter)
        sampler._ts.fill_(ts_value)
        sampler._capture()
    g = sampler._graphs[mode]
    times = []
    for _ in range(reps + 1):
        sampler._x.copy_(x)
        sampler._t.fill_(counter)
        sampler._ts.fill_(ts_value)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize(dev)
        times.append(e0.elapsed_time(e1))
    slot = 4 * (int(counter) & 3)                 # the replayed step ran with this counter: its slot of the probe ring
    walked, total = int(sampler._probe_host[slot]), int(sampler._probe_host[slot + 1])
    return statistics.median(times[1:]), (walked / total if total else None)


def _interp_mean(points, lo, hi):
    """Mean over the integers lo .. hi of the piecewise-linear interpolant through `points` [(x, y)]."""
    import numpy as np
    pts = sorted(points)
    xs, ys = [p[0] for p in pts], [p[1] for p in pts]
    return float(np.interp(np.arange(lo, hi + 1), xs, ys).mean())


def _leg(sampler, B, x0, dh, grid, counter_of, net_time_of, abar_t_of, reps):
    rows = []
    for key in grid:
        x = q_sample(x0, abar_t_of(key), dh, seed=11)
        row = {"t": key}
        for mode in ("once", "whole"):
            ms, share = _time_replays(sampler, mode, x, counter_of(key), net_time_of(key), reps)
            row[mode + "_ms"] = round(ms, 4)
            if mode == "once":
                row["tiles_walked_frac"] = None if share is None else round(share, 4)
        share = row["tiles_walked_frac"]
        row["adaptive"] = "whole" if (share is not None and share > sampler.WHOLE_ABOVE) else "once"
        row["adaptive_ms"] = row[row["adaptive"] + "_ms"]
        rows.append(row)
    return rows


def measure(build_sampler, build_fast_sampler, build_refine, device, B, dh, reps=4):
    """-> the `trajectory` object of the bench line.  build_sampler() -> GraphedReverseSampler (adaptive),
    build_fast_sampler() -> GraphedFastSampler (adaptive, S = 50 quadratic), build_refine() -> (refine_net, callable)."""
    from point_diffusion_refinement_amd.pointnet2 import generation as G
    x0, cond, label = synthetic_surface_batch(B, seed=0, device=device)
    out = {"input": "x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) eps, x_0 = %d synthetic tori in [-0.5, 0.5]^3 (configs."
                    "synthetic_surface_batch), condition = their mirrored partial views; one captured step per form, "
                    "median of %d replays, walked share as published by the step; `adaptive` = the form the sampler's "
                    "rule (walked share > WHOLE_ABOVE -> whole) replays" % (B, reps)}
    # ---- DDPM, T = 1000
    s = build_sampler()
    s.begin((B, 2048, 3), cond, label, x_T=q_sample(x0, s.T - 1, dh, seed=11))
    out["whole_above"] = s.WHOLE_ABOVE
    rows = _leg(s, B, x0, dh, T_GRID, lambda t: t, lambda t: float(t), lambda t: t, reps)
    T = s.T
    means = {k: _interp_mean([(r["t"], r[k]) for r in rows], 0, T - 1) for k in ("once_ms", "whole_ms", "adaptive_ms")}
    out["ddpm_t1000"] = {"grid": rows, "weighted_mean_ms": {k[:-3]: round(v, 4) for k, v in means.items()},
                         "weighted_mean_cloud_steps_per_s": round(B / means["adaptive_ms"] * 1e3, 1),
                         "note": "piecewise-linear in t over t = 0 .. %d, one weight per step" % (T - 1)}
    dense = rows[-1]
    out["dense_input"] = {"t": 0, "tiles_walked_frac": dense["tiles_walked_frac"],
                          "dedup_on_ms": dense["once_ms"], "dedup_off_ms": dense["whole_ms"],
                          "product_default_ms": dense["adaptive_ms"], "product_default_form": dense["adaptive"],
                          "product_default_over_whole": round(dense["adaptive_ms"] / dense["whole_ms"], 4)}
    del s
    torch.cuda.empty_cache()
    # ---- FastDPM S = 50 (quadratic): counter i <-> network time tau[i]
    f = build_fast_sampler()
    tau = [float(v) for v in f.f_tau.tolist()]
    S = len(tau)
    f.begin((B, 2048, 3), cond, label, x_T=q_sample(x0, int(round(tau[S - 1])), dh, seed=11))
    pick = sorted(set(i for i in (0, 1, 2, 3, 5, 7, 10, 14, 19, 25, 32, 40, S - 1) if i < S))
    rows = _leg(f, B, x0, dh, pick, lambda i: i, lambda i: tau[i], lambda i: min(max(int(round(tau[i])), 0), dh["T"] - 1),
                reps)
    for r in rows:
        r["tau"] = round(tau[r["t"]], 2)
        r["step"] = r.pop("t")
    means = {k: _interp_mean([(r["step"], r[k]) for r in rows], 0, S - 1) for k in ("once_ms", "whole_ms", "adaptive_ms")}
    out["fastdpm_s50"] = {"grid": rows, "mean_ms_per_step": {k[:-3]: round(v, 4) for k, v in means.items()},
                          "loop_s": {k[:-3]: round(v * S / 1e3, 4) for k, v in means.items()},
                          "note": "quadratic tau list of util_fastdpmv2 (configs[4]); piecewise-linear over the step index"}
    del f
    torch.cuda.empty_cache()
    # ---- refinement forward (+ x8 upsampling) on the finished surface, both forms, eager
    refine_net = build_refine()
    res = {}
    for name, on in (("dedup_off", False), ("dedup_on", True)):
        def run():
            refine_net.reset_cond_features()
            saved, refine_net.dedup = refine_net.dedup, on
            try:
                return refine_net(x0, cond, ts=None, label=label)
            finally:
                refine_net.dedup = saved
        with torch.no_grad():
            run()
            torch.cuda.synchronize(device)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                run()
                torch.cuda.synchronize(device)
                ts.append((time.perf_counter() - t0) * 1e3)
        res[name + "_ms"] = round(statistics.median(ts), 3)
    with torch.no_grad():
        t0 = time.perf_counter()
        G.refine_completion(refine_net, x0, cond, label, 0.001, 8)
        torch.cuda.synchronize(device)
        res["product_default_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    res["note"] = "one eager forward of the refinement network (fresh condition branch) on the tori themselves; " \
                  "generation.refine_completion (the product default, incl. x8 upsampling) turns the deduplication off"
    out["refinement_forward"] = res
    return out
