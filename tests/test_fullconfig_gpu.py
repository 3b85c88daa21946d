"""BASELINE configs at their REAL sizes on one GPU.

  * configs[4] (FastDPM S = 50 'var' / 'quadratic' / kappa 0.5 on the DDPM architecture, refinement forward with
    x8 point upsampling to N = 16384, Chamfer at 16384^2):
        util_fastdpmv2.py:307-381, 455-476; completion_eval.py:159-168; models/point_upsample_module.py:4-28;
        chamfer_loss_new.py:234-245 of the reference.
  * configs[1] (B = 32, N = 2048, T = 1000 DDPM reverse sampling, util.py:184-255): the fused + hipGraph path
    against the layer-by-layer HIP path over ONE CPU noise stream.  Two fp32 implementations with different
    summation orders cannot agree element-wise over 1000 steps (discrete decisions -- FPS picks, ball membership,
    ReLU / mask boundaries -- flip at near-ties); what is asserted is what SURVEY section 7
    calls distributional parity: how many clouds diverged, Chamfer / EMD between the two outputs of every cloud,
    and agreement of the job's summary metrics.

Measured values of every quantity asserted here are also written to gpurun_out/fullconfig_*.json (diagnostics).
"""
import contextlib
import io
import json
import os

import numpy as np
import pytest
import torch

from oracle import pdr_oracle as O   # checker
from point_diffusion_refinement_amd.pointnet2 import generation as G
from point_diffusion_refinement_amd.pointnet2 import util
from point_diffusion_refinement_amd.pointnet2 import util_fastdpmv2 as F
from point_diffusion_refinement_amd.pointnet2.chamfer_loss_new import calc_cd
from point_diffusion_refinement_amd.pointnet2.configs import (DIFFUSION_CONFIG, ddpm_pointnet_config,
                                                              refinement_pointnet_config, synthetic_batch)
from point_diffusion_refinement_amd.pointnet2.emd import earth_mover_distance
from point_diffusion_refinement_amd.pointnet2.fused_network import FusedCloudConditionNet
from point_diffusion_refinement_amd.pointnet2.models.pointnet2_with_pcld_condition import PointNet2CloudCondition
from point_diffusion_refinement_amd.pointnet2.reverse_sampler import GraphedFastSampler, GraphedReverseSampler
from point_diffusion_refinement_amd.pointnet2_ops import _ext

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _diag(name, payload):
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "fullconfig_%s.json" % name), "w") as f:
            json.dump(payload, f, indent=1)
    except OSError:
        pass


def _rel(a, b):
    return ((a - b).abs() / (b.abs() + 1.0)).flatten(1)


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def test_config5_fastdpm_refine_upsample_chamfer_full_size(cuda):
    """configs[4] as ONE pipeline at B = 2 (FastDPM S = 50 -> refinement + x8 -> Chamfer at 16384 points): the graph-captured
    fused sampler and the fused refinement network against the layer-by-layer composition of the SAME HIP ops -- a
    consistency test of the fusion / graph capture with flip-tolerant bounds, NOT the parity evidence.  Parity of both
    stages against the reference's own outputs at this size is pinned by reference-generated fixtures in
    tests/test_reference_golden.py: test_full_ddpm_config_fastdpm_s50_matches_reference (every one of the 50 x_t,
    teacher-forced: 1.2e-6) and test_full_refinement_config_forward_and_x8_upsampling_match_reference (2.1e-7)."""
    torch.manual_seed(0)
    net = PointNet2CloudCondition(ddpm_pointnet_config()).eval().to(cuda)            # shipped architecture, random init
    fused = FusedCloudConditionNet(net)
    dh = util.calc_diffusion_hyperparams(**DIFFUSION_CONFIG)
    B, S = 2, 50
    _, cond, label = synthetic_batch(B, seed=11, device=cuda)

    # ---- (1) S = 50 FastDPM: graph-captured fused loop vs the reference-style eager loop, one CPU noise stream
    util.set_device(cuda)
    util.set_noise_source('cpu')
    try:
        torch.manual_seed(501)
        want = _quiet(F.fast_sampling_function_v2, net, (B, 2048, 3), dh, DIFFUSION_CONFIG, length=S,
                      sampling_method='var', schedule='quadratic', kappa=0.5, label=label, verbose=False,
                      condition=cond)
    finally:
        util.set_device(None)
    sampler = GraphedFastSampler(fused, dh, DIFFUSION_CONFIG, length=S, sampling_method='var', schedule='quadratic',
                                 kappa=0.5, noise='cpu', use_graph=True)
    torch.manual_seed(501)
    got = sampler.sample((B, 2048, 3), cond, label)
    assert got.shape == (B, 2048, 3) and bool(torch.isfinite(got).all())
    rel = _rel(got, want)
    cd_p, cd_t = calc_cd(got, want)
    # scale of the comparison: mean nearest-neighbour spacing inside the reference output
    d2, _, _ = _ext.knn_points(want.contiguous(), want.contiguous(), 2)
    spacing = d2[..., 1].sqrt().mean(1)
    diag = {"median_rel": rel.median(1).values.tolist(), "max_rel": rel.max(1).values.tolist(),
            "cd_p_between": cd_p.tolist(), "nn_spacing": spacing.tolist()}
    _diag("config5_fastdpm", diag)
    # measured (MI355X, r2): one of the two clouds carries a flipped near-tie (median 4e-4, max 0.29, cd_p 0.6 % of
    # the spacing), the other agrees to 2e-7; bounds leave room for different flips after kernel changes
    assert (rel.median(1).values < 5e-3).all(), diag          # the bulk of every cloud agrees
    assert float(rel.max()) < 2.0, diag                        # a flipped near-tie moves single points, bounded
    assert (cd_p < 0.05 * spacing).all(), diag                 # as point SETS: far below the point spacing

    # ---- (2) refinement forward + x8 upsampling to (B, 16384, 3): fused vs layer-by-layer network
    torch.manual_seed(1)
    rnet = PointNet2CloudCondition(refinement_pointnet_config(8)).eval().to(cuda)
    rfused = FusedCloudConditionNet(rnet)
    # (a random-init DDPM leaves clouds of scale ~30; the refinement stage sees completed shapes in [-1, 1]^3)
    coarse = (got / got.abs().amax(dim=(1, 2), keepdim=True)).contiguous()
    with torch.no_grad():
        rnet.reset_cond_features()
        disp_ref = rnet(coarse, cond, ts=None, label=label)                      # (B, 2048, 3 (f + 1)) displacement
        disp = rfused(coarse, cond, ts=None, label=label)
        fine_ref = G.refine_completion(rnet, coarse, cond, label, 0.001, 8)
        fine = G.refine_completion(rfused, coarse, cond, label, 0.001, 8)
    assert disp.shape == disp_ref.shape == (B, 2048, 3 * (8 + 1))               # centre + 8 offsets per point
    assert fine.shape == (B, 16384, 3) and fine_ref.shape == (B, 16384, 3)
    err = (disp - disp_ref).abs() / (disp_ref.abs() + 1.0)
    d5 = {"disp_err_max": float(err.max()), "disp_err_frac_below_1e-3": float((err < 1e-3).float().mean()),
          "coord_abs_max": float((fine - fine_ref).abs().max()), "disp_abs_mean": float(disp_ref.abs().mean())}
    _diag("config5_refine", d5)
    assert err.max() < 1e-2 and (err < 1e-3).float().mean() > 0.99, d5          # the network-level bar
    assert float(disp_ref.abs().mean()) > 1e-3, d5                               # a non-degenerate comparison
    assert (fine - fine_ref).abs().max() < 1e-4, d5                              # refined coordinates

    # ---- (3) Chamfer at 16384^2: oracle on one cloud (values + indices), float64 brute force on a slice
    g = torch.Generator().manual_seed(9)
    gt = (torch.rand(B, 16384, 3, generator=g) - 0.5).to(cuda)
    out = (fine / 2).contiguous()
    cd_p, cd_t, f1 = calc_cd(out, gt, calc_f1=True, f1_threshold=1e-4)
    assert cd_p.shape == (B,) and bool(torch.isfinite(cd_t).all())
    dx, ix, dy, iy = O.chamfer(gt[:1].cpu().numpy(), out[:1].cpu().numpy())     # chamfer_distance(gt, output)
    gd, gi, _ = _ext.knn_points(gt[:1].contiguous(), out[:1].contiguous(), 1)
    hd, hi, _ = _ext.knn_points(out[:1].contiguous(), gt[:1].contiguous(), 1)
    assert np.array_equal(gi.cpu().numpy()[..., 0], ix) and np.array_equal(hi.cpu().numpy()[..., 0], iy)
    assert np.array_equal(gd.cpu().numpy()[..., 0], dx) and np.array_equal(hd.cpu().numpy()[..., 0], dy)
    np.testing.assert_allclose(cd_t[:1].cpu().numpy(), dx.mean(1) + dy.mean(1), rtol=1e-5)
    np.testing.assert_allclose(cd_p[:1].cpu().numpy(), (np.sqrt(dx).mean(1) + np.sqrt(dy).mean(1)) / 2, rtol=1e-5)
    q = gt[1, :1024].double()
    brute = ((q[:, None, :] - out[1].double()[None, :, :]) ** 2).sum(-1).min(1).values
    kd, _, _ = _ext.knn_points(gt[1:2, :1024].contiguous(), out[1:2].contiguous(), 1)
    np.testing.assert_allclose(kd[0, :, 0].cpu().numpy(), brute.cpu().numpy(), rtol=1e-4, atol=1e-9)


def test_config2_b32_t1000_fused_hip_graph_vs_layerwise_hip_distributional_parity(cuda):
    """BASELINE configs[1] at its real size, B = 32, T = 1000.  HIP vs HIP (said so in the name): the
    fused hipGraph sampler -- both captured forms and the per-step switch -- against the layer-by-layer loop over the same
    native ops; the reference itself cannot run 32,000 network calls on the CPU in a test, its full-size parity is the
    T = 6 / T = 3 / S = 50 / dense goldens of tests/test_reference_golden.py."""
    torch.manual_seed(0)
    net = PointNet2CloudCondition(ddpm_pointnet_config()).eval().to(cuda)
    fused = FusedCloudConditionNet(net)
    dh = util.calc_diffusion_hyperparams(**DIFFUSION_CONFIG)
    B, N = 32, 2048
    _, cond, label = synthetic_batch(B, seed=21, device=cuda)

    torch.manual_seed(900)
    a = GraphedReverseSampler(fused, dh, noise='cpu', use_graph=True).sample((B, N, 3), cond, label)
    torch.manual_seed(900)
    b = GraphedReverseSampler(net, dh, noise='cpu', use_graph=False).sample((B, N, 3), cond, label)
    # the same recursion with eps == 0 (closed form of x_T and the noise draws): |a - x0| is what the 1000
    # network evaluations CONTRIBUTED to the result; differences are judged against that, not against |x|
    torch.manual_seed(900)
    A, Ab, Sg = dh["Alpha"], dh["Alpha_bar"], dh["Sigma"]
    x0 = torch.normal(0, 1, size=(B, N, 3))
    for t in range(999, -1, -1):
        x0 = x0 / torch.sqrt(A[t])
        if t > 0:
            x0 = x0 + Sg[t] * torch.normal(0, 1, size=(B, N, 3))
    x0 = x0.to(cuda)
    assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all())

    contrib = (b - x0).abs().flatten(1).mean(1)                       # per cloud: mean |network contribution|
    diff = (a - b).abs().flatten(1)
    rel_contrib = diff.median(1).values / contrib                      # median |a-b| relative to that
    rel = _rel(a, b)
    diverged = rel_contrib > 1e-2                                      # a flipped discrete decision somewhere
    cd_p, cd_t = calc_cd(a, b)
    emd = earth_mover_distance(a, b)
    d2, _, _ = _ext.knn_points(b.contiguous(), b.contiguous(), 2)
    spacing = d2[..., 1].sqrt().mean(1)
    g = torch.Generator().manual_seed(33)
    gt = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(cuda)
    _, rec_a = G.evaluate_batch(lambda c, l: a, None, label, gt)
    _, rec_b = G.evaluate_batch(lambda c, l: b, None, label, gt)
    sa, sb = G.summarize(rec_a), G.summarize(rec_b)
    diag = {"clouds": B, "diverged": int(diverged.sum()), "rel_contrib_median": rel_contrib.tolist(),
            "median_rel_x": rel.median(1).values.tolist(), "max_rel_x": rel.max(1).values.tolist(),
            "contrib_mean_abs": contrib.tolist(), "x_abs_mean": float(b.abs().mean()),
            "cd_p_between": cd_p.tolist(), "emd_between": emd.tolist(), "nn_spacing": spacing.tolist(),
            "summary_fused": sa, "summary_layerwise": sb}
    _diag("config2_t1000", diag)
    # measured (MI355X, r2): 0 of 32 clouds diverged; per-cloud median |a-b|/(|b|+1) <= 1.7e-3; Chamfer between the
    # two outputs <= 1.8 % of the point spacing; summary metrics equal to 2e-5 relative
    assert int(diverged.sum()) <= B // 4, diag                         # most clouds never flip
    assert (rel.median(1).values < 1e-2).all(), diag                   # every cloud: the bulk of x agrees
    assert (cd_p < 0.1 * spacing).all(), diag                          # as point sets: far below the spacing
    for k in ("avg_cd", "avg_cd_p", "avg_emd"):
        assert abs(sa[k] - sb[k]) <= 1e-3 * abs(sb[k]), (k, sa[k], sb[k])
    assert abs(sa["avg_f1"] - sb["avg_f1"]) <= 1e-3
