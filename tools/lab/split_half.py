"""Accuracy of the split (f16 hi + lo, three MFMAs per product) arithmetic, per layer and end to end.

    python -m tools.lab.split_half > gpurun_out/split_half.json

Per layer: max |y - y64| / (|x| . |w|) over all outputs, at input magnitudes 1, 1e-2 and 1e-4 (the lo parts of the
small inputs are SUBNORMAL halves -- the figure tells whether the MFMA honours them: it does).  The same script on
the bf16 hi / lo build this mode used before (commit aa89a5f + the type switch) is the comparison quoted in
DESIGN.md section 4.2.  Network: eps of the
shipped DDPM architecture vs the exact fused network.  Sampler: T = 6 and T = 30 graph-captured loops vs the
reference-style loop, per-cloud max / median of |d| / max(|want|, rms)."""
import contextlib
import io
import json
import sys

import torch

from point_diffusion_refinement_amd import _lib
from point_diffusion_refinement_amd.pointnet2 import fused_network as FN
from point_diffusion_refinement_amd.pointnet2 import util
from point_diffusion_refinement_amd.pointnet2.configs import ddpm_pointnet_config, synthetic_batch
from point_diffusion_refinement_amd.pointnet2.models.pointnet2_with_pcld_condition import PointNet2CloudCondition
from point_diffusion_refinement_amd.pointnet2.reverse_sampler import GraphedReverseSampler


def conv_of(W, bias):
    m = torch.nn.Conv2d(W.shape[1], W.shape[0], 1).to(W.device)
    m.weight.data = W.reshape(W.shape[0], W.shape[1], 1, 1).clone()
    m.bias.data = bias.clone()
    return FN.Conv([m])


def layer_error(dev, P, Cin, Cout, rpb, magnitude):
    g = torch.Generator().manual_seed(Cin + Cout)
    B = P // rpb
    x = (torch.randn(P, Cin, generator=g) * magnitude).to(dev)
    W = (torch.randn(Cout, Cin, generator=g) / Cin ** 0.5).to(dev)
    conv = conv_of(W, torch.zeros(Cout, device=dev))
    act = FN.Act([(x, 0, Cin, Cin, 1)], P, B, rpb)
    ref = x.double() @ W.t().double()
    bound = x.abs().double() @ W.t().abs().double()
    lib = _lib.load()
    Y = torch.empty((P, FN._ldy(Cout)), device=dev)
    tm = lib.pdr_fused_layer_tile_rows(rpb, Cout)
    part = torch.empty((B * ((rpb + tm - 1) // tm), Cout, 2), device=dev)
    FN._PRECISION[0] = "split_f16"
    ok = FN._run_layer_split(lib, act, conv, act.struct(), Y.data_ptr(), Y.shape[1], part, Cout)
    FN._PRECISION[0] = "f32"
    assert ok, "split path not taken"
    split = float(((Y[:, :Cout].double() - ref).abs() / bound).max())
    Ye, _, _ = FN.run_layer(act, conv, stats=True)
    exact = float(((Ye[:, :Cout].double() - ref).abs() / bound).max())
    return split, exact


def rel(got, want):
    rms = want.flatten(1).pow(2).mean(1).sqrt().view(-1, 1, 1)
    return ((got - want).abs() / torch.maximum(want.abs(), rms)).flatten(1)


def main():
    dev = torch.device("cuda:0")
    out = {"half": "f16", "layers": [], "network": {}, "sampler": {}}
    for P, Cin, Cout, rpb in ((1 << 16, 256, 256, 8192), (4096, 256, 64, 1024), (8192, 512, 512, 1024)):
        for mag in (1.0, 1e-2, 1e-4):
            s, e = layer_error(dev, P, Cin, Cout, rpb, mag)
            out["layers"].append({"shape": [P, Cin, Cout], "magnitude": mag, "split_max": s, "exact_max": e})
    torch.manual_seed(0)
    net = PointNet2CloudCondition(ddpm_pointnet_config()).eval().to(dev)
    exact = FN.FusedCloudConditionNet(net)
    split = FN.FusedCloudConditionNet(net, precision="split_f16")
    x, cond, label = synthetic_batch(2, seed=3, device=dev)
    ts = torch.tensor([500.0, 20.0], device=dev)
    with torch.no_grad():
        net.reset_cond_features()
        net(x, cond, ts=ts, label=label, use_retained_condition_feature=True)
        x2 = x * 0.9
        ref = net(x2, cond, ts=ts - 1, label=label, use_retained_condition_feature=True)
        exact.sync_condition()
        a = exact(x2, cond, ts=ts - 1, label=label, use_retained_condition_feature=True).clone()
        b = split(x2, cond, ts=ts - 1, label=label, use_retained_condition_feature=True).clone()
    for name, got, want in (("split_vs_exact_fused", b, a), ("split_vs_layerwise", b, ref), ("exact_vs_layerwise", a, ref)):
        r = rel(got, want)
        out["network"][name] = {"max": float(r.max()), "median": float(r.median())}
    util.set_noise_source('cpu')
    for T in (6, 30):
        dh = util.calc_diffusion_hyperparams(T, 1e-4, 0.02)
        util.set_device(dev)
        torch.manual_seed(77)
        with contextlib.redirect_stdout(io.StringIO()):
            want = util.sampling(net, (2, 2048, 3), dh, label=label, verbose=False, condition=cond)
        util.set_device(None)
        row = {}
        for name, f in (("split", split), ("exact", exact)):
            torch.manual_seed(77)
            got = GraphedReverseSampler(f, dh, noise='cpu', use_graph=True).sample((2, 2048, 3), cond, label)
            r = rel(got, want)
            row[name] = {"per_cloud_max": [float(v) for v in r.max(1).values],
                         "per_cloud_median": [float(v) for v in r.median(1).values]}
        out["sampler"]["T%d" % T] = row
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
