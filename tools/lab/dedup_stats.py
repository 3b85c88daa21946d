"""valid-tile fraction of every deduplicated block at the bench input (lab)"""
import torch
import bench
from point_diffusion_refinement_amd.pointnet2 import fused_network as FN
from point_diffusion_refinement_amd.pointnet2.configs import synthetic_batch


def main():
    dev = torch.device("cuda", 0)
    plans = []
    init = FN.Dedup.__init__

    def rec(self, *a, **k):
        init(self, *a, **k)
        plans.append(self)
    FN.Dedup.__init__ = rec
    smp, _ = bench.build_sampler(dev, False, True, "f32")
    x_T, cond, label = synthetic_batch(32, bench.N_POINTS, bench.M_COND, seed=0, device=dev)
    smp.begin((32, bench.N_POINTS, 3), cond, label, x_T=x_T)
    plans.clear()
    smp.advance(1)
    torch.cuda.synchronize()
    kept = tot = 0
    for p in plans:
        n, t = int(p.n_tiles), p.B * p.tpb
        deg = float((p.row_w > 0).float().mean())
        print("m=%5d K=%d: %6d of %6d tiles walked (%.0f%%); queries in skipped tiles %.0f%%" % (p.m, p.K, n, t, 100.0 * n / t, 100 * deg))
        kept += n
        tot += t
    print("all plans: %.0f%% of the tiles walked" % (100.0 * kept / tot))


if __name__ == "__main__":
    main()
