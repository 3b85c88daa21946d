"""Op-level roofline rows at the BASELINE sizes (B = 32 clouds of the DDPM configuration; 1000-pair batches for the
evaluation ops), HIP-event timed on the launch stream.  One JSON object on stdout.

For every op: algorithmic pair evaluations x 8 flop -> achieved fp32 VALU TFLOP/s against the 157.3 TF
vector peak, and algorithmic bytes (inputs + outputs once) -> achieved HBM GB/s against 8 TB/s.  These ops are VALU- /
latency-bound by construction (150-750 flop per byte vs a machine balance of ~20): both fractions are reported, the
larger one names the binding roof."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from point_diffusion_refinement_amd.pointnet2 import emd  # noqa: E402
from point_diffusion_refinement_amd.pointnet2.chamfer_loss_new import calc_cd  # noqa: E402
from point_diffusion_refinement_amd.pointnet2_ops import _ext  # noqa: E402

VALU_PEAK, HBM_PEAK = 157.3e12, 8.0e12


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def row(name, secs, pairs, byt, note=""):
    fl = 8.0 * pairs
    r = {"op": name, "us": round(secs * 1e6, 1), "pair_evals": int(pairs), "algorithmic_bytes": int(byt),
         "valu_TFLOPs": round(fl / secs / 1e12, 3), "valu_frac_of_157TF": round(fl / secs / VALU_PEAK, 4),
         "hbm_GBps": round(byt / secs / 1e9, 2), "hbm_frac_of_8TBps": round(byt / secs / HBM_PEAK, 5)}
    r["binding"] = "valu" if r["valu_frac_of_157TF"] >= r["hbm_frac_of_8TBps"] else "hbm"
    if note:
        r["note"] = note
    return r


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    B = 32
    rows = []
    x = torch.randn(B, 2048, 3, generator=g).to(dev)
    cond = (torch.rand(B, 3072, 3, generator=g) * 2 - 1).to(dev)
    # FPS chain of one step: 2048 -> 1024 -> 256 -> 64 -> 16
    levels = [x]
    for n, m in ((2048, 1024), (1024, 256), (256, 64), (64, 16)):
        src = levels[-1]
        s = timed(lambda: _ext.furthest_point_sampling(src, m))
        rows.append(row("furthest_point_sampling %d->%d (B=32)" % (n, m), s, B * n * (m - 1), B * (12 * n + 4 * m),
                        "latency-bound: %d dependent rounds = %.0f ns per round" % (m - 1, s / (m - 1) * 1e9)))
        idx = _ext.furthest_point_sampling(src, m)
        levels.append(torch.gather(src, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous())
    for (q, p, r, qn, pn) in ((x, cond, 0.1, 2048, 3072), (levels[1], x, 0.1, 1024, 2048),
                              (levels[1], levels[1], 0.2, 1024, 1024), (levels[2], levels[1], 0.2, 256, 1024)):
        s = timed(lambda: _ext.ball_query(q, p, r, 32))
        rows.append(row("ball_query %dx%d r=%.1f ns=32 (B=32)" % (qn, pn, r), s, B * qn * pn,
                        B * (12 * (qn + pn) + 4 * qn * 32 + 4 * qn)))
    for (a, b, an, bn) in ((x, levels[1], 2048, 1024), (levels[1], levels[2], 1024, 256)):
        s = timed(lambda: _ext.knn_points(a, b, 8))
        rows.append(row("knn_points K=8 %dx%d (B=32)" % (an, bn), s, B * an * bn, B * (12 * (an + bn) + an * 8 * 12)))
    nb = 1000
    p1 = (torch.rand(nb, 2048, 3, generator=g) - 0.5).to(dev)
    p2 = (torch.rand(nb, 2048, 3, generator=g) - 0.5).to(dev)
    s = timed(lambda: calc_cd(p1, p2, calc_f1=True), reps=5)
    rows.append(row("Chamfer + F1, 1000 pairs of 2048^2", s, nb * 2 * 2048 * 2048, nb * (2 * 2048 * 12 + 2 * 2048 * 12)))
    s = timed(lambda: emd.earth_mover_distance(p1, p2), reps=3)
    r = row("approx. EMD (cost only), 1000 pairs of 2048^2", s, nb * 30 * 2048 * 2048, nb * (2 * 2048 * 12 + 4),
            "per pair-eval: distance (8 flop) + one v_exp_f32 (quarter rate) -> transcendental-bound")
    r["texp_per_s"] = round(nb * 30 * 2048 * 2048 / s / 1e12, 3)
    rows.append(r)
    big1 = (torch.rand(8, 16384, 3, generator=g) - 0.5).to(dev)
    big2 = (torch.rand(8, 16384, 3, generator=g) - 0.5).to(dev)
    s = timed(lambda: calc_cd(big1, big2), reps=3)
    rows.append(row("Chamfer, 8 pairs of 16384^2 (config 5)", s, 8 * 2 * 16384 * 16384, 8 * 4 * 16384 * 12))
    print(json.dumps({"peaks": {"fp32_valu_TFLOPs": 157.3, "hbm_TBps": 8.0}, "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
