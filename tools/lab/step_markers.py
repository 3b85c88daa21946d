"""Time stamps INSIDE an untraced, graph-replayed reverse step (VERDICT r2 'prove or remove the start-up stall without a
tracer'): pdr_mark_time kernels (one thread, 100 MHz wall clock) are captured at the fork, at the waits of the main
stream and after every block; the stamps of R replays are read back and reported relative to the step's first stamp.
    python -m tools.lab.step_markers [out.json]      (MARK_DETAIL=1: stamps inside every grouped block, program order)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import build_sampler  # noqa: E402
from point_diffusion_refinement_amd.pointnet2 import fused_network as FN  # noqa: E402
from point_diffusion_refinement_amd.pointnet2.configs import synthetic_batch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B = int(os.environ.get("MARK_B", "32"))
    sampler, _ = build_sampler(dev, use_graph=True, precision=os.environ.get("MARK_PRECISION", "f32"))
    x_T, cond, label = synthetic_batch(B, seed=0, device=dev)
    buf = torch.zeros(512, dtype=torch.int64, device=dev)
    FN.MARKS = {"buf": buf, "names": [], "detail": os.environ.get("MARK_DETAIL", "0") == "1"}
    sampler.begin((B, 2048, 3), cond, label, x_T=x_T)       # eager first step (marks fire, ignored)
    FN.MARKS["names"] = []
    sampler.advance(1)                                       # warm-up on a side stream + capture: marks fire TWICE
    names = FN.MARKS["names"]
    half = len(names) // 2
    assert names[:half] == names[half:], names               # warm-up pass then captured pass: same sequence
    # the captured launches wrote to slots [half, 2 half)
    FN_DETAIL = FN.MARKS["detail"]
    FN.MARKS = None
    R = 20
    rows = []
    torch.cuda.synchronize()
    # MARK_BACK_TO_BACK=n: n replays queued without a synchronisation in between, the stamps of the LAST one read -- the
    # steady state of a sampling loop, where the host submits step i + 1 while step i runs (a lone replay, the default,
    # also shows how long the submission of the step's own nodes takes: the two pictures differ at the head of the step)
    b2b = int(os.environ.get("MARK_BACK_TO_BACK", "1"))
    for _ in range(R):
        sampler.advance(b2b)
        torch.cuda.synchronize()
        rows.append(buf[half:2 * half].cpu().numpy().astype("int64"))
    import numpy as np
    t = np.stack(rows)                                       # (R, marks) in 10 ns ticks
    rel = (t - t[:, :1]) * 0.01                              # us after step:begin
    med = np.median(rel, axis=0)
    order = np.argsort(med, kind="stable") if not FN_DETAIL else np.arange(len(med))
    out = {"batch": B, "replays": R, "back_to_back": b2b, "precision": os.environ.get("MARK_PRECISION", "f32"),
           "unit": "us after step:begin (median of replays; 100 MHz clock)",
           "marks": [{"name": names[i], "median_us": float(med[i]), "min_us": float(rel[:, i].min()),
                      "max_us": float(rel[:, i].max())} for i in order]}
    # wall time of a replay by events for reference
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        sampler.advance(1)
    e1.record()
    torch.cuda.synchronize()
    out["ms_per_step_with_marks"] = e0.elapsed_time(e1) / 10
    for m in out["marks"]:
        print("%9.1f us  %s" % (m["median_us"], m["name"]))
    print("ms/step (with the marker launches in the graph): %.3f" % out["ms_per_step_with_marks"])
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
