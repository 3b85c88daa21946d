"""BASELINE configs[2]: Chamfer + F1 + approximate EMD on 10k synthetic (2048,3) cloud pairs, 1 GPU.

    python tools/eval_bench.py [--pairs 10000] [--batch 1000]

Prints pairs/s for calc_cd (two K=1 nearest-neighbour searches, cd_p / cd_t / F1) and for the
cost-only EMD path (pdr_emd_cost), the algorithmic rates behind them, and checks a few pairs against
the CPU oracle (indices bit-exact, values 1e-4)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import pdr_oracle as O  # noqa: E402  (checker only)
from point_diffusion_refinement_amd.pointnet2 import emd  # noqa: E402
from point_diffusion_refinement_amd.pointnet2.chamfer_loss_new import calc_cd  # noqa: E402
from point_diffusion_refinement_amd.pointnet2_ops import _ext  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=10000)
    ap.add_argument("--batch", type=int, default=1000)
    ap.add_argument("--points", type=int, default=2048)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    n = args.points
    a = (torch.rand(args.batch, n, 3, generator=g) - 0.5).to(dev)     # the /2/scale range of the harness
    b = (torch.rand(args.batch, n, 3, generator=g) - 0.5).to(dev)
    # parity on a slice
    cd_p, cd_t, f1 = calc_cd(a[:4], b[:4], calc_f1=True)
    dx, ix, dy, iy = O.chamfer(b[:4].cpu().numpy(), a[:4].cpu().numpy())
    _, gi, _ = _ext.knn_points(b[:4].contiguous(), a[:4].contiguous(), 1)
    assert np.array_equal(gi.cpu().numpy()[..., 0], ix)
    np.testing.assert_allclose(cd_t.cpu().numpy(), dx.mean(1) + dy.mean(1), rtol=1e-5)
    e = emd.earth_mover_distance(a[:2], b[:2]).cpu().numpy()
    np.testing.assert_allclose(e, O.emd(a[:2].cpu().numpy(), b[:2].cpu().numpy()), rtol=1e-4)

    reps = max(1, args.pairs // args.batch)
    out = {}
    for name, fn in (("chamfer_f1", lambda: calc_cd(a, b, calc_f1=True)),
                     ("emd_cost", lambda: emd.earth_mover_distance(a, b))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name + "_pairs_per_s"] = round(reps * args.batch / dt, 1)
        out[name + "_ms_per_pair"] = round(dt / (reps * args.batch) * 1e3, 4)
    # algorithmic work per pair: Chamfer 8.39 M pair-evals x 8 flop; EMD 125.8 M exp + distance evals
    out["chamfer_gflops"] = round(out["chamfer_f1_pairs_per_s"] * 2 * n * n * 8 / 1e9, 1)
    out["emd_gexp_per_s"] = round(out["emd_cost_pairs_per_s"] * 30 * n * n / 1e9, 1)
    out["pairs"], out["points"] = reps * args.batch, n
    print(json.dumps(out))


if __name__ == "__main__":
    main()
