"""Second MLP of the feature-propagation blocks (64- / 256-point levels, B = 32) as ONE pdr_point_chain launch vs the
layer launches (conv + fold + conv + fold + activation), each alone on the chip: event-timed, hipGraph replay of 20
calls.   python -m tools.lab.chain_time"""
import torch

from point_diffusion_refinement_amd.pointnet2 import fused_network as FN
from point_diffusion_refinement_amd.pointnet2.configs import ddpm_pointnet_config
from point_diffusion_refinement_amd.pointnet2.models.pointnet2_with_pcld_condition import PointNet2CloudCondition


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = PointNet2CloudCondition(ddpm_pointnet_config()).eval().to(dev)
    fused = FN.FusedCloudConditionNet(net)
    B = 32
    bank = fused.bank
    g = torch.Generator(device=dev).manual_seed(1)
    for k, W in bank.W.items():
        bank.out[k] = torch.randn(B, W.shape[0], device=dev, generator=g)
    for name, fp, n in (("fp4 (64 points)", fused.fp[-1], 64), ("fp3 (256 points)", fused.fp[-2], 256),
                        ("fp2 (1024 points)", fused.fp[-3], 1024)):
        mlp = fp.mlp2
        Cin = mlp.first.Cin
        D = fp.att.D
        Cs = Cin - D - 3
        interp = torch.randn(B * n, D, device=dev, generator=g)
        feats = torch.randn(B, n, Cs, device=dev, generator=g)
        xyz = torch.randn(B, n, 3, device=dev, generator=g)

        def x2():
            return FN.Act([(interp, 0, D, D, 1), (FN.xyz4(feats), 0, Cs, FN._pad4(Cs), 1), (FN.xyz4(xyz), 0, 3, 4, 1)],
                          B * n, B, n)

        def chain():
            return mlp.chain(x2(), bank)

        def layers():
            h2, _, _, _ = mlp(x2(), bank, relu_stats_extra=False)
            return FN.materialize(h2)
        res = {}
        for label, fn in (("chain", chain), ("layers", layers)):
            with torch.no_grad():
                out = fn()
                if out is None:
                    res[label] = None
                    continue
                torch.cuda.synchronize()
                gph = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    fn()
                torch.cuda.current_stream().wait_stream(s)
                with torch.cuda.graph(gph):
                    for _ in range(20):
                        out = fn()
                for _ in range(3):
                    gph.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ts = []
                for _ in range(10):
                    e0.record()
                    gph.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
                res[label] = (sorted(ts)[len(ts) // 2], out.clone())
        a, b = res["chain"], res["layers"]
        if a is None:
            print("%-18s Cin %d -> %d: chain unsupported, layers %.1f us" % (name, Cin, mlp.Clast, b[0]))
        else:
            err = float(((a[1] - b[1]).abs() / (b[1].abs() + 1)).max())
            print("%-18s Cin %d -> %d: chain %.1f us, layer launches %.1f us, max rel diff %.2e" %
                  (name, Cin, mlp.Clast, a[0], b[0], err))


if __name__ == "__main__":
    main()
