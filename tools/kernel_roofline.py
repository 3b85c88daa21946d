"""Live roofline of the dominant hand-written kernel of the reverse step (used by bench.py).

The dominant kernel (largest share of the step in profiles/*_kernel_stats.csv) is
`fused_layer_ws_kernel<2,2,2,2,32,false,false>`: the wave-specialised 128 x 128-tile fp32-MFMA layer kernel
(csrc/fused_layer_ws.hip) that evaluates the wide 1x1-conv GEMMs of the SA / feature-transfer /
kNN-FP blocks.  Its roof is the dense fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md).

Every launch of that instantiation inside a reverse step is bracketed with HIP events recorded on
the stream the kernel is launched on (torch's current stream; the ops read the same stream handle),
over `reps` eager repetitions of the step:
    achieved = (sum of algorithmic flops of those launches) / (sum of their durations)
           = (average algorithmic flops per launch) / (average launch duration)
with algorithmic flops of a launch = 2 * P * Cin * Cout (P positions, no padding counted).
"""
import json
import os

import torch

from point_diffusion_refinement_amd import _lib
from point_diffusion_refinement_amd.pointnet2 import fused_network as FN

DOMINANT_VARIANT = 4          # pdr_fused_layer_variant(): 128 x 128 tile, 2-D grid
DOMINANT_SYMBOL = "fused_layer_ws_kernel<2, 2, 2, 2, 32, false, false>"
FP32_MFMA_PEAK_TFLOPS = 157.3


TRAFFIC_FILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                            "r1_pmc_traffic.json")


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/pmc_traffic.sh: FETCH_SIZE
    and WRITE_SIZE in separate rocprofv3 runs of this bench, FETCH_SIZE doubled per the gfx950 correction), or None."""
    try:
        for k in json.load(open(TRAFFIC_FILE))["kernels"]:
            if DOMINANT_SYMBOL in k["kernel"]:
                return k["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def dominant_kernel_roofline(sampler, reps=3):
    lib = _lib.load()
    records = []
    original = FN.run_layer

    def timed(act, conv, *a, **k):
        # the plain (no residual, no gathered source) 128 x 128 instantiation
        hit = (lib.pdr_fused_layer_variant(act.rpb, conv.Cout) == DOMINANT_VARIANT and act.radd is None
               and act.gidx is None)
        if not hit:
            return original(act, conv, *a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = original(act, conv, *a, **k)
        e1.record()
        seg_bytes = sum(4 * sg[2] * act.P // sg[4] for sg in act.segs)
        records.append((e0, e1, 2.0 * act.P * conv.Cin * conv.Cout, seg_bytes + 4.0 * act.P * conv.Cout))
        return out

    FN.run_layer = timed
    try:
        with torch.no_grad():
            for _ in range(reps):
                sampler._step()           # eager: the graph is not involved
        torch.cuda.synchronize()
    finally:
        FN.run_layer = original
    ms = sum(a.elapsed_time(b) for a, b, _, _ in records)
    flops = sum(r[2] for r in records)
    byts = sum(r[3] for r in records)
    n = len(records)
    achieved = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": measured_traffic(),
            "kernel": DOMINANT_SYMBOL, "launches_per_step": n // reps,
            "avg_launch_us": round(ms / n * 1e3, 2), "avg_gflop_per_launch": round(flops / n / 1e9, 3),
            "algorithmic_GBps": round(byts / (ms * 1e-3) / 1e9, 1),
            "note": "fp32 MFMA (v_mfma_f32_32x32x2_f32) roof; events on the launch stream, eager step x%d" % reps}
