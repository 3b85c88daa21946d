"""Parity of the HIP kernels (through the C ABI, via the `_ext` shim) against the CPU oracle.

Bar (BASELINE.json north_star): indices bit-exact for FPS / ball_query / kNN / three_nn /
group / gather; 1e-4 relative for EMD, Chamfer and interpolation values.
"""
import numpy as np
import pytest
import torch

from oracle import pdr_oracle as O
from point_diffusion_refinement_amd.pointnet2 import emd
from point_diffusion_refinement_amd.pointnet2.chamfer_loss_new import calc_cd, chamfer_distance
from point_diffusion_refinement_amd.pointnet2_ops import _ext
from point_diffusion_refinement_amd.pointnet2_ops import pointnet2_utils as PU

pytestmark = pytest.mark.gpu


def rng(seed=0):
    return np.random.default_rng(seed)


def dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def host(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ FPS
@pytest.mark.parametrize("B,N,m", [(1, 1, 1), (2, 16, 16), (3, 64, 16), (2, 100, 37), (4, 256, 64), (2, 1000, 333),
                                   (4, 1024, 256), (3, 2048, 1024), (2, 3072, 1024), (1, 4096, 2048),
                                   (1, 5000, 100), (1, 12288, 64), (1, 20000, 40)])
def test_fps_index_exact(cuda, B, N, m):
    x = rng(N + m).uniform(-1, 1, (B, N, 3)).astype(np.float32)
    got = host(_ext.furthest_point_sampling(dev(x, cuda), m))
    assert np.array_equal(got, O.furthest_point_sampling(x, m))


def test_fps_ties_duplicates_and_origin_exclusion(cuda):
    g = np.stack(np.meshgrid(np.arange(6), np.arange(6), np.arange(6), indexing="ij"), -1).reshape(-1, 3)
    p = np.concatenate([g, g[10:110], g[::3]]).astype(np.float32)[None]          # 388 pts, many exact ties
    p = np.concatenate([p, p[:, ::-1]], 0).copy()
    for m in (2, 50, 200):
        assert np.array_equal(host(_ext.furthest_point_sampling(dev(p, cuda), m)), O.furthest_point_sampling(p, m))
    q = np.full((1, 128, 3), 0.0, np.float32)
    q[0, :, 0] = 1
    q[0, 0, 0] = 3
    assert host(_ext.furthest_point_sampling(dev(q, cuda), 2))[0, 1] == 64        # bit-reversed thread order
    tiny = np.full((2, 300, 3), 0.01, np.float32)                                # every point excluded
    assert (host(_ext.furthest_point_sampling(dev(tiny, cuda), 9)) == 0).all()
    mixed = rng(5).uniform(-0.05, 0.05, (2, 700, 3)).astype(np.float32)          # many |p|^2 <= 1e-3
    assert np.array_equal(host(_ext.furthest_point_sampling(dev(mixed, cuda), 64)),
                          O.furthest_point_sampling(mixed, 64))


def test_fps_is_permutation_prefix_at_full_size(cuda):
    """BASELINE size (B=32, 2048 -> 1024): size-independent properties + oracle on a slice."""
    x = torch.randn(32, 2048, 3, generator=torch.Generator().manual_seed(1))
    idx = _ext.furthest_point_sampling(x.to(cuda), 1024).cpu()
    assert (idx[:, 0] == 0).all()
    assert all(len(set(r.tolist())) == 1024 for r in idx)                        # no repeats
    assert np.array_equal(idx[:3].numpy(), O.furthest_point_sampling(x[:3].numpy(), 1024))


# ----------------------------------------------------------- ball query
@pytest.mark.parametrize("B,n,m,r,ns", [(1, 1, 1, 0.5, 4), (2, 16, 16, 1.6, 32), (2, 64, 16, 0.8, 32),
                                        (3, 100, 37, 0.3, 8), (2, 256, 256, 0.4, 32), (2, 1024, 1024, 0.2, 32),
                                        (2, 2048, 1024, 0.1, 32), (2, 3072, 2048, 0.1, 32), (1, 4096, 100, 0.15, 64),
                                        (1, 5000, 77, 0.2, 100), (2, 300, 50, 0.001, 16)])
def test_ball_query_index_exact(cuda, B, n, m, r, ns):
    rr = rng(n * 7 + m)
    p = rr.uniform(-1, 1, (B, n, 3)).astype(np.float32)
    q = rr.uniform(-1, 1, (B, m, 3)).astype(np.float32)
    if m <= n:
        q[:, : m // 2] = p[:, : m // 2]                                          # queries that ARE cloud points
    gi, gc = _ext.ball_query(dev(q, cuda), dev(p, cuda), r, ns)
    oi, oc = O.ball_query(q, p, r, ns)
    assert np.array_equal(host(gc), oc)
    assert np.array_equal(host(gi), oi)


def test_ball_query_through_python_signature(cuda):
    """pointnet2_utils.ball_query(radius, nsample, xyz, new_xyz) swaps to the native (new_xyz, xyz) order."""
    p = rng(3).uniform(-1, 1, (2, 500, 3)).astype(np.float32)
    q = p[:, :100].copy()
    gi, gc = PU.ball_query(0.3, 16, dev(p, cuda), dev(q, cuda))
    oi, oc = O.ball_query(q, p, 0.3, 16)
    assert np.array_equal(host(gi), oi) and np.array_equal(host(gc), oc) and (oc >= 1).all()


# ------------------------------------------------------ gather / group
@pytest.mark.parametrize("B,C,N,m", [(1, 1, 1, 1), (2, 3, 100, 37), (2, 35, 2048, 1024), (3, 320, 64, 16)])
def test_gather_and_grad(cuda, B, C, N, m):
    rr = rng(C + N)
    f = rr.standard_normal((B, C, N)).astype(np.float32)
    idx = rr.integers(0, N, (B, m)).astype(np.int32)
    assert np.array_equal(host(_ext.gather_points(dev(f, cuda), dev(idx, cuda))), O.gather_points(f, idx))
    g = rr.standard_normal((B, C, m)).astype(np.float32)
    np.testing.assert_allclose(host(_ext.gather_points_grad(dev(g, cuda), dev(idx, cuda), N)),
                               O.gather_points_grad(g, idx, N), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,C,N,np_,ns", [(1, 1, 1, 1, 1), (2, 4, 100, 37, 8), (2, 32, 3072, 2048, 32),
                                          (2, 131, 256, 64, 32), (3, 9, 16, 16, 32)])
def test_group_and_grad(cuda, B, C, N, np_, ns):
    rr = rng(C + N + ns)
    f = rr.standard_normal((B, C, N)).astype(np.float32)
    idx = rr.integers(0, N, (B, np_, ns)).astype(np.int32)
    assert np.array_equal(host(_ext.group_points(dev(f, cuda), dev(idx, cuda))), O.group_points(f, idx))
    g = rr.standard_normal((B, C, np_, ns)).astype(np.float32)
    np.testing.assert_allclose(host(_ext.group_points_grad(dev(g, cuda), dev(idx, cuda), N)),
                               O.group_points_grad(g, idx, N), rtol=1e-4, atol=1e-4)


# -------------------------------------------------------- three_nn/interp
@pytest.mark.parametrize("B,n,m", [(1, 1, 1), (2, 64, 16), (2, 256, 64), (2, 2048, 1024), (1, 100, 2), (2, 333, 1500)])
def test_three_nn_and_interpolate(cuda, B, n, m):
    rr = rng(n + m)
    u = rr.uniform(-1, 1, (B, n, 3)).astype(np.float32)
    k = rr.uniform(-1, 1, (B, m, 3)).astype(np.float32)
    d2, idx = _ext.three_nn(dev(u, cuda), dev(k, cuda))
    od2, oidx = O.three_nn(u, k)
    assert np.array_equal(host(idx), oidx) and np.array_equal(host(d2), od2)
    dist, idx2 = PU.three_nn(dev(u, cuda), dev(k, cuda))                          # Python surface returns sqrt
    np.testing.assert_array_equal(host(dist), np.sqrt(od2))
    if m >= 3:
        feats = rr.standard_normal((B, 19, m)).astype(np.float32)
        w = rr.random((B, n, 3)).astype(np.float32)
        got = host(_ext.three_interpolate(dev(feats, cuda), idx, dev(w, cuda)))
        assert np.array_equal(got, O.three_interpolate(feats, oidx, w))
        g = rr.standard_normal((B, 19, n)).astype(np.float32)
        np.testing.assert_allclose(host(_ext.three_interpolate_grad(dev(g, cuda), idx, dev(w, cuda), m)),
                                   O.three_interpolate_grad(g, oidx, w, m), rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------ kNN
@pytest.mark.parametrize("B,n1,n2,K", [(1, 1, 1, 1), (2, 64, 16, 8), (2, 256, 64, 8), (2, 1024, 256, 8),
                                       (2, 2048, 1024, 8), (2, 3072, 1024, 8), (2, 40, 5, 8), (2, 500, 300, 3),
                                       (1, 300, 2500, 16), (1, 100, 100, 32), (2, 2048, 2048, 1)])
def test_knn_exact(cuda, B, n1, n2, K):
    rr = rng(n1 + 3 * n2 + K)
    x = rr.uniform(-1, 1, (B, n1, 3)).astype(np.float32)
    y = rr.uniform(-1, 1, (B, n2, 3)).astype(np.float32)
    if n1 <= n2:
        x[:, ::2] = y[:, : (n1 + 1) // 2]                                         # exact zero distances
    d, i, nn = _ext.knn_points(dev(x, cuda), dev(y, cuda), K, return_nn=True)
    od, oi = O.knn(x, y, K)
    assert np.array_equal(host(i), oi) and np.array_equal(host(d), od)
    gathered = np.take_along_axis(y[:, None].repeat(n1, 1), np.maximum(oi, 0)[..., None].repeat(3, -1), 2)
    gathered[oi < 0] = 0
    assert np.array_equal(host(nn), gathered)


@pytest.mark.parametrize("n2,K", [(64, 8), (100, 8), (256, 8), (700, 5), (1024, 8), (1000, 2)])
def test_knn_wave_kernel_ties_and_partial_chunks(cuda, n2, K):
    """The wave-per-query kernel (K <= 8, 64 <= n2 <= 1024) on inputs built to stress its selection: an integer
    lattice (many exactly equal distances -> ranking by lower index), a cloud of identical points (every point is a
    candidate: > 64 candidates -> the exact extraction fallback), duplicated points, cloud sizes that leave the
    last 64-point chunk partial; bit-equal to the oracle and to pdr_knn_group."""
    rr = rng(900 + n2 + K)
    lattice = rr.integers(-3, 4, (2, n2, 3)).astype(np.float32)
    same = np.broadcast_to(np.array([0.25, -0.5, 0.125], np.float32), (2, n2, 3)).copy()
    dup = rr.uniform(-1, 1, (2, n2, 3)).astype(np.float32)
    dup[:, n2 // 2:] = dup[:, : n2 - n2 // 2]
    for y in (lattice, same, dup):
        x = np.concatenate([y[:, :40], rr.integers(-3, 4, (2, 60, 3)).astype(np.float32)], 1)
        d, i, _ = _ext.knn_points(dev(x, cuda), dev(y, cuda), K)
        od, oi = O.knn(x, y, K)
        assert np.array_equal(host(i), oi) and np.array_equal(host(d), od)
        dg, ig, wg = _ext.knn_group(dev(x, cuda), dev(y, cuda), K)
        assert np.array_equal(host(ig), oi.astype(np.int32)) and np.array_equal(host(dg), od)
        recip = 1.0 / (od.astype(np.float64) + 1e-8)
        np.testing.assert_allclose(host(wg), recip / recip.sum(-1, keepdims=True), rtol=2e-6)


@pytest.mark.parametrize("B,n1,n2,K", [(2, 300, 257, 1), (2, 128, 64, 8), (1, 10, 6, 8), (3, 2048, 2048, 1)])
def test_knn_grad_vs_oracle(cuda, B, n1, n2, K):
    """pdr_knn_points_grad through the C ABI == oracle (atomics: 1e-5 relative to the gradient scale)."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((B, n1, 3)).astype(np.float32)
    y = rng.standard_normal((B, n2, 3)).astype(np.float32)
    g = rng.standard_normal((B, n1, K)).astype(np.float32)
    xt = torch.tensor(x, device=cuda, requires_grad=True)
    yt = torch.tensor(y, device=cuda, requires_grad=True)
    d, idx, _ = _ext.knn_points(xt, yt, K)
    assert d.requires_grad and not idx.requires_grad
    (d * torch.tensor(g, device=cuda)).sum().backward()
    _, oi = O.knn(x, y, K)
    np.testing.assert_array_equal(idx.cpu().numpy(), oi)
    gx, gy = O.knn_grad(x, y, oi, g)
    scale = max(1.0, float(np.abs(gy).max()))
    np.testing.assert_allclose(xt.grad.cpu().numpy(), gx, rtol=1e-5, atol=1e-5 * scale)
    np.testing.assert_allclose(yt.grad.cpu().numpy(), gy, rtol=1e-5, atol=1e-5 * scale)


def test_calc_cd_is_differentiable_like_the_reference_loss(cuda):
    """train.py:518-533 back-propagates calc_cd; gradient == float64 brute-force Chamfer autograd."""
    rng = np.random.default_rng(12)
    out = rng.uniform(-0.5, 0.5, (2, 256, 3)).astype(np.float32)
    gt = rng.uniform(-0.5, 0.5, (2, 300, 3)).astype(np.float32)
    o = torch.tensor(out, device=cuda, requires_grad=True)
    cd_p, cd_t = calc_cd(o, torch.tensor(gt, device=cuda))
    (cd_t.sum() + 0.5 * cd_p.sum()).backward()
    o64 = torch.tensor(out, dtype=torch.float64, requires_grad=True)
    g64 = torch.tensor(gt, dtype=torch.float64)
    dmat = ((g64.unsqueeze(2) - o64.unsqueeze(1)) ** 2).sum(-1)          # (B, n_gt, n_out)
    d1, d2 = dmat.min(2).values, dmat.min(1).values
    ref_t = d1.mean(1) + d2.mean(1)
    ref_p = (d1.sqrt().mean(1) + d2.sqrt().mean(1)) / 2
    (ref_t.sum() + 0.5 * ref_p.sum()).backward()
    np.testing.assert_allclose(cd_t.detach().cpu().numpy(), ref_t.detach().numpy(), rtol=1e-4)
    np.testing.assert_allclose(o.grad.cpu().numpy(), o64.grad.numpy(), rtol=2e-4, atol=1e-6)


def test_group_knn_layout(cuda):
    rr = rng(11)
    x = rr.uniform(-1, 1, (2, 64, 3)).astype(np.float32)
    y = rr.uniform(-1, 1, (2, 16, 3)).astype(np.float32)
    f = rr.standard_normal((2, 5, 16)).astype(np.float32)
    out = host(PU.group_knn(dev(x, cuda), dev(y, cuda), dev(f, cuda), 8, transpose=True))
    assert out.shape == (2, 5 + 11, 64, 8)
    od, oi = O.knn(x, y, 8)
    np.testing.assert_array_equal(out[:, 5], od)                                  # squared distances channel
    np.testing.assert_allclose(out[:, 6].sum(-1), 1.0, rtol=1e-5)                 # normalised weights
    np.testing.assert_array_equal(out[:, 13:16], np.broadcast_to(x.transpose(0, 2, 1)[..., None], (2, 3, 64, 8)))


@pytest.mark.parametrize("B,n1,n2", [(1, 1, 1), (2, 100, 200), (3, 2048, 2048), (2, 1025, 1031), (1, 5000, 3), (2, 7, 4100),
                                     (1, 16384, 16384)])
def test_chamfer_nn_both_directions_bit_equal_to_knn(cuda, B, n1, n2):
    """pdr_chamfer_nn (packed-math K = 1 kernel, both directions in one launch) == oracle knn(K=1) bit for bit:
    distances AND first-minimum-wins indices, with exact ties (integer lattice) and exact zero distances."""
    rr = rng(n1 * 7 + n2)
    x = rr.uniform(-1, 1, (B, n1, 3)).astype(np.float32)
    y = rr.uniform(-1, 1, (B, n2, 3)).astype(np.float32)
    # many exact ties: snap half of both clouds to a coarse lattice; exact zeros: copy some points across
    x[:, ::2] = np.round(x[:, ::2] * 4) / 4
    y[:, ::2] = np.round(y[:, ::2] * 4) / 4
    k = min(n1, n2) // 3
    if k:
        x[:, 1:1 + k] = y[:, :k]
    dx, ix, dy, iy = (host(t) for t in _ext.chamfer_nn(dev(x, cuda), dev(y, cuda)))
    od, oi = O.knn(x, y, 1)
    rd, ri = O.knn(y, x, 1)
    assert np.array_equal(ix, oi[..., 0]) and np.array_equal(dx, od[..., 0])
    assert np.array_equal(iy, ri[..., 0]) and np.array_equal(dy, rd[..., 0])
    # the K = 1 route of knn_points is the same kernel (single direction)
    d1, i1, _ = _ext.knn_points(dev(x, cuda), dev(y, cuda), 1)
    assert np.array_equal(host(i1)[..., 0], ix) and np.array_equal(host(d1)[..., 0], dx)


@pytest.mark.parametrize("B,n1,n2,K", [(2, 64, 16, 8), (2, 2048, 1024, 8), (1, 300, 40, 3), (2, 1024, 256, 16)])
def test_knn_group_indices_and_weights(cuda, B, n1, n2, K):
    """pdr_knn_group = knn_points (index-exact, int32) + group_knn's normalised 1/(d2+1e-8) weights."""
    rr = rng(n1 + n2 + K)
    x = rr.uniform(-1, 1, (B, n1, 3)).astype(np.float32)
    y = rr.uniform(-1, 1, (B, n2, 3)).astype(np.float32)
    x[:, ::5] = y[:, : len(x[0, ::5])] if n2 >= len(x[0, ::5]) else x[:, ::5]       # exact zero distances
    d, i, w = _ext.knn_group(dev(x, cuda), dev(y, cuda), K)
    od, oi = O.knn(x, y, K)
    assert i.dtype == torch.int32
    assert np.array_equal(host(i), oi.astype(np.int32)) and np.array_equal(host(d), od)
    recip = 1.0 / (od.astype(np.float64) + 1e-8)
    np.testing.assert_allclose(host(w), recip / recip.sum(-1, keepdims=True), rtol=2e-6)


# -------------------------------------------------------------- Chamfer
def test_chamfer_unit_test_protocol(cuda):
    """Same protocol as the reference's ChamferDistancePytorch/unit_test.py:22-33."""
    rr = rng(2)
    x = rr.random((4, 100, 3)).astype(np.float32)
    y = rr.random((4, 200, 3)).astype(np.float32)
    cx, cy, _ = chamfer_distance(dev(x, cuda), dev(y, cuda), batch_reduction=None, point_reduction=None)
    d = ((x[:, :, None].astype(np.float64) - y[:, None].astype(np.float64)) ** 2).sum(-1)
    assert np.mean((host(cx) - d.min(2)) ** 2) < 1e-8 and np.mean((host(cy) - d.min(1)) ** 2) < 1e-8
    _, ix, _ = _ext.knn_points(dev(x, cuda), dev(y, cuda), 1)
    assert np.array_equal(host(ix)[..., 0], d.argmin(2))


def test_calc_cd_and_f1_vs_oracle(cuda):
    rr = rng(4)
    out = rr.uniform(-0.5, 0.5, (3, 2048, 3)).astype(np.float32)
    gt = (out + rr.normal(0, 0.01, out.shape)).astype(np.float32)
    cd_p, cd_t, f1 = calc_cd(dev(out, cuda), dev(gt, cuda), calc_f1=True)
    d1, _, d2, _ = O.chamfer(gt, out)                                             # chamfer_distance(gt, output)
    np.testing.assert_allclose(host(cd_t), d1.mean(1) + d2.mean(1), rtol=1e-5)
    np.testing.assert_allclose(host(cd_p), (np.sqrt(d1).mean(1) + np.sqrt(d2).mean(1)) / 2, rtol=1e-5)
    p1, p2 = (d1 < 1e-4).mean(1), (d2 < 1e-4).mean(1)
    np.testing.assert_allclose(host(f1), 2 * p1 * p2 / (p1 + p2), rtol=1e-5)
    # symmetry property: swapping the clouds swaps the two directed terms, cd_t unchanged
    _, cd_t2 = calc_cd(dev(gt, cuda), dev(out, cuda))
    np.testing.assert_allclose(host(cd_t2), host(cd_t), rtol=1e-6)


# ------------------------------------------------------------------ EMD
def test_emd_two_point_known_answer(cuda):
    p1 = np.array([[[1.7, -0.1, 0.1], [0.1, 1.2, 0.3]]], dtype=np.float32).repeat(3, 0)
    p2 = np.array([[[0.3, 1.8, 0.2], [1.2, -0.2, 0.3]]], dtype=np.float32).repeat(3, 0)
    cost, match = emd.earth_mover_distance(dev(p1, cuda), dev(p2, cuda), return_match=True)
    np.testing.assert_allclose(host(match)[0], [[0, 1], [1, 0]], atol=1e-6)
    np.testing.assert_allclose(host(cost), 0.355, rtol=1e-4)
    np.testing.assert_allclose(host(emd.earth_mover_distance(dev(p1, cuda), dev(p2, cuda))), 0.355, rtol=1e-4)


@pytest.mark.parametrize("B,n,m", [(2, 64, 64), (3, 300, 300), (2, 256, 512), (2, 512, 256), (2, 1000, 1000),
                                   (2, 2048, 2048)])
def test_emd_vs_oracle(cuda, B, n, m):
    rr = rng(n + m)
    a = rr.uniform(-0.5, 0.5, (B, n, 3)).astype(np.float32)
    b = rr.uniform(-0.5, 0.5, (B, m, 3)).astype(np.float32)
    omatch = O.approxmatch(a, b)
    ocost = O.matchcost(a, b, omatch)
    cost, match = emd.earth_mover_distance(dev(a, cuda), dev(b, cuda), return_match=True)
    fused = emd.earth_mover_distance(dev(a, cuda), dev(b, cuda))
    np.testing.assert_allclose(host(cost), ocost / max(n, m), rtol=1e-4)
    np.testing.assert_allclose(host(fused), ocost / max(n, m), rtol=1e-4)
    # single entries of the soft assignment amplify the __expf (v_exp_f32) vs expf difference
    # through 10 multiplicative levels; the transport marginals and the cost stay at 1e-4
    np.testing.assert_allclose(host(match), omatch, rtol=2e-2, atol=5e-4)
    np.testing.assert_allclose(host(match).sum(1), omatch.sum(1), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(host(match).sum(2), omatch.sum(2), rtol=1e-4, atol=1e-5)
    # matchcost on a GIVEN match (the 3rd pybind symbol) and its gradients
    np.testing.assert_allclose(host(emd.matchcost_forward(dev(a, cuda), dev(b, cuda), dev(omatch, cuda))), ocost,
                               rtol=1e-4)
    g = rr.random(B).astype(np.float32)
    g1, g2 = emd.matchcost_backward(dev(g, cuda), dev(a, cuda), dev(b, cuda), dev(omatch, cuda))
    o1, o2 = O.matchcost_grad(g, a, b, omatch)
    np.testing.assert_allclose(host(g1), o1, rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(host(g2), o2, rtol=1e-3, atol=1e-5)


def test_emd_autograd_path(cuda):
    rr = rng(1)
    a = dev(rr.uniform(-0.5, 0.5, (2, 128, 3)).astype(np.float32), cuda).requires_grad_(True)
    b = dev(rr.uniform(-0.5, 0.5, (2, 128, 3)).astype(np.float32), cuda).requires_grad_(True)
    c = emd.earth_mover_distance(a, b)
    c.sum().backward()
    assert a.grad.shape == a.shape and torch.isfinite(a.grad).all() and a.grad.abs().sum() > 0


# -------------------------------------------------- error behaviour / API
def test_reference_error_behaviour(cuda):
    x = torch.rand(2, 64, 3, device=cuda)
    with pytest.raises(RuntimeError, match="contiguous"):
        _ext.furthest_point_sampling(x.transpose(0, 1), 4)
    with pytest.raises(RuntimeError, match="float"):
        _ext.furthest_point_sampling(x.double(), 4)
    with pytest.raises(RuntimeError, match="int"):
        _ext.gather_points(x.transpose(1, 2).contiguous(), torch.zeros(2, 4, dtype=torch.int64, device=cuda))


def test_ops_honour_the_current_stream_and_graph_capture(cuda):
    """Every op launches on torch's current stream with no hidden sync/alloc => hipGraph-capturable."""
    x = torch.rand(4, 512, 3, device=cuda)
    ref = _ext.furthest_point_sampling(x, 64).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            _ext.furthest_point_sampling(x, 64)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            idx = _ext.furthest_point_sampling(x, 64)
            bi, bc = _ext.ball_query(x[:, :64].contiguous(), x, 0.2, 16)
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(idx, ref)
    oi, oc = O.ball_query(host(x[:, :64]), host(x), 0.2, 16)
    assert np.array_equal(host(bi), oi) and np.array_equal(host(bc), oc)


# ------------------------------------------------------------ reverse-step update
@pytest.mark.parametrize("mode", [0, 1])
def test_reverse_update_is_bit_identical_to_the_torch_expression(cuda, mode):
    """pdr_reverse_update == the reference's elementwise expressions evaluated op by op in PyTorch
    (util.py:246-250 / util_fastdpmv2.py:186-204), step index read from device memory, strided eps rows."""
    from point_diffusion_refinement_amd import _lib
    g = torch.Generator().manual_seed(5 + mode)
    B, N, T = 3, 1000, 50
    x = torch.randn(B, N, 3, generator=g).to(cuda)
    eps4 = torch.randn(B, N, 4, generator=g).to(cuda)
    eps = eps4[:, :, :3]                                                   # leading dimension 4
    z = torch.randn(B, N, 3, generator=g).to(cuda)
    a, b, c = (torch.rand(T, generator=g).to(cuda) + 0.5 for _ in range(3))
    for step in (T - 1, 7, 0):
        t = torch.tensor([step], dtype=torch.int64, device=cuda)
        A, Bc, C = a[step], b[step], c[step]
        want = (x - A * eps) / Bc + C * z if mode == 0 else (x * A) + (Bc * eps + C * z)
        got = x.clone()
        _lib.check(_lib.load().pdr_reverse_update(got.data_ptr(), eps.data_ptr(), eps.stride(1), z.data_ptr(),
                                                  a.data_ptr(), b.data_ptr(), c.data_ptr(), t.data_ptr(), B * N, mode,
                                                  torch.cuda.current_stream().cuda_stream), "reverse_update")
        assert torch.equal(got, want), (mode, step, float((got - want).abs().max()))


@pytest.mark.parametrize("mode", [0, 1])
def test_reverse_step_update_bookkeeping_and_noise(cuda, mode):
    """pdr_reverse_step: with an explicit z the update is bit-identical to the torch expression AND the launch
    decrements the device step counter and publishes the next network time input (float(t-1) or tau[t-1]); with
    in-kernel noise the normals are N(0,1) (moments, no correlation between neighbours / between draws), reproducible
    from (key, draw number), different for another key or draw, and the draw number advances by one per launch."""
    from point_diffusion_refinement_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(50 + mode)
    B, N, T = 3, 1001, 50                                                  # 9009 elements: a partial last quad
    x = torch.randn(B, N, 3, generator=g).to(cuda)
    eps4 = torch.randn(B, N, 4, generator=g).to(cuda)
    eps = eps4[:, :, :3]
    z = torch.randn(B, N, 3, generator=g).to(cuda)
    a, b, c = (torch.rand(T, generator=g).to(cuda) + 0.5 for _ in range(3))
    tau = (torch.arange(T, dtype=torch.float32) * 1.75 + 0.125).to(cuda)
    st = torch.cuda.current_stream().cuda_stream
    ticket = torch.zeros(1, dtype=torch.int32, device=cuda)
    acc = torch.zeros(2, dtype=torch.int32, device=cuda)
    published = torch.zeros(16, dtype=torch.int32).pin_memory()            # pinned host memory: the 4-slot probe ring
    for step, table in ((T - 1, None), (7, tau), (0, None)):
        t = torch.tensor([step], dtype=torch.int64, device=cuda)
        ts = torch.full((1,), -5.0, device=cuda)
        A, Bc, C = a[step], b[step], c[step]
        want = (x - A * eps) / Bc + C * z if mode == 0 else (x * A) + (Bc * eps + C * z)
        got = x.clone()
        acc.copy_(torch.tensor([17 + step, 40], dtype=torch.int32))
        _lib.check(lib.pdr_reverse_step(got.data_ptr(), eps.data_ptr(), eps.stride(1), z.data_ptr(), a.data_ptr(),
                                        b.data_ptr(), c.data_ptr(), t.data_ptr(), None if table is None else table.data_ptr(),
                                        ts.data_ptr(), None, ticket.data_ptr(), B * N, mode, acc.data_ptr(),
                                        published.data_ptr(), st), "reverse_step")
        assert torch.equal(got, want), (mode, step, float((got - want).abs().max()))
        assert int(t) == step - 1 and int(ticket) == 0
        # the neighbourhood probe of the step: published to the (host-visible) slot and reset by the same last workgroup
        torch.cuda.synchronize()
        slot = 4 * (step & 3)                                  # slot t & 3 = {walked, tiles, step counter, written}
        assert published[slot:slot + 4].tolist() == [17 + step, 40, step, 1] and acc.tolist() == [0, 0]
        expect_ts = -5.0 if step == 0 else (float(tau[step - 1]) if table is not None else float(step - 1))
        assert float(ts) == expect_ts
    # in-kernel noise: isolate z through x = 0, eps = 0, coefficients (a, b, c) = (1, 1, 1)
    Bn, Nn = 32, 2048
    one = torch.ones(4, device=cuda)
    zero_eps = torch.zeros(Bn, Nn, 3, device=cuda)

    def draw(key, number):
        xs = torch.zeros(Bn, Nn, 3, device=cuda)
        t = torch.tensor([3], dtype=torch.int64, device=cuda)
        rng = torch.tensor([key, number], dtype=torch.int64, device=cuda)
        _lib.check(lib.pdr_reverse_step(xs.data_ptr(), zero_eps.data_ptr(), 3, None, one.data_ptr(), one.data_ptr(),
                                        one.data_ptr(), t.data_ptr(), None, None, rng.data_ptr(), ticket.data_ptr(),
                                        Bn * Nn, mode, None, None, st), "reverse_step")
        assert rng.tolist() == [key, number + 1] and int(t) == 2
        return xs.flatten().double().cpu()
    z0, z0b, z1, zk = draw(1234567, 0), draw(1234567, 0), draw(1234567, 1), draw(-99, 0)
    assert torch.equal(z0, z0b) and not torch.equal(z0, z1) and not torch.equal(z0, zk)
    n = z0.numel()                                                         # 196,608 samples
    for s in (z0, z1, zk):
        assert abs(float(s.mean())) < 4 / n ** 0.5 and abs(float(s.var()) - 1) < 4 * (2 / n) ** 0.5
        assert abs(float((s ** 3).mean())) < 4 * (15 / n) ** 0.5 and abs(float((s ** 4).mean()) - 3) < 4 * (96 / n) ** 0.5
        assert abs(float((s[1:] * s[:-1]).mean())) < 4 / n ** 0.5         # neighbours (same Philox block / adjacent blocks)
        assert float(s.abs().max()) < 6.5 and float((s.abs() > 3).double().mean()) > 0.0015
    assert abs(float((z0 * z1).mean())) < 4 / n ** 0.5 and abs(float((z0 * zk).mean())) < 4 / n ** 0.5
    # against the normal CDF at a few quantiles
    for qv, p in ((-1.0, 0.158655), (0.0, 0.5), (0.5, 0.691462), (2.0, 0.977250)):
        assert abs(float((z0 < qv).double().mean()) - p) < 4 * (p * (1 - p) / n) ** 0.5
