"""Size-independent properties of the HIP ops at BASELINE.json's full sizes (B = 32 clouds, 3072-point
condition, 2048-point clouds, 10k-pair evaluation batches), where the CPU oracle would take minutes.
Float64 brute force on the GPU is the judge; queries whose decision lies within 1e-5 (relative) of a boundary
are excluded from set-equality checks (the fp32 rounding of the kernels' fused expression decides those, and
that rounding is pinned bit-exactly against the oracle at smaller sizes in test_ops_gpu.py)."""
import numpy as np
import pytest
import torch

from point_diffusion_refinement_amd.pointnet2 import emd
from point_diffusion_refinement_amd.pointnet2.chamfer_loss_new import calc_cd
from point_diffusion_refinement_amd.pointnet2_ops import _ext

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cuda():
    return torch.device("cuda:0")


def _clouds(B, n, seed, cuda, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, n, 3, generator=g) * (hi - lo) + lo).to(cuda)


def test_ball_query_full_batch_properties(cuda):
    B, n, m, r, ns = 32, 3072, 2048, 0.1, 32
    xyz, new = _clouds(B, n, 1, cuda), _clouds(B, m, 2, cuda)
    idx, cnt = _ext.ball_query(new, xyz, r, ns)
    assert idx.shape == (B, m, ns) and cnt.shape == (B, m)
    d2 = torch.cdist(new.double(), xyz.double()) ** 2                      # (B, m, n)
    inside = d2 < r * r
    borderline = ((d2 - r * r).abs() < 1e-5 * r * r).any(-1)
    want = inside.sum(-1).clamp(max=ns)
    ok = ~borderline
    assert bool((cnt[ok] == want[ok]).all())
    # every reported neighbour is inside the ball, ascending, and equals the FIRST `count` inside points
    k = torch.arange(ns, device=cuda).view(1, 1, ns)
    valid = k < cnt.unsqueeze(-1)
    got_d = d2.gather(2, idx.long())
    assert bool((got_d[valid] < r * r * (1 + 1e-5)).all())
    asc = (idx[..., 1:] > idx[..., :-1]) | ~valid[..., 1:]
    assert bool(asc.all())
    first = torch.where(inside, torch.arange(n, device=cuda).view(1, 1, n), n).sort(-1).values[..., :ns]
    same = (idx.long() == first) | ~valid
    assert bool(same[ok].all())
    # padding = the first hit; empty balls: count 0, row all zero
    pad_ok = (idx == idx[..., :1]) | valid
    assert bool(pad_ok.all())
    assert bool((idx[cnt == 0] == 0).all())


def test_knn_full_batch_properties(cuda):
    B, n1, n2, K = 32, 2048, 1024, 8
    x, y = _clouds(B, n1, 3, cuda), _clouds(B, n2, 4, cuda)
    d, idx, nn = _ext.knn_points(x, y, K, return_nn=True)
    assert bool((d[..., 1:] >= d[..., :-1]).all())                          # ascending
    near = y.gather(1, idx.reshape(B, -1, 1).expand(-1, -1, 3)).reshape(B, n1, K, 3)
    assert torch.equal(nn, near)                                            # nn is the gather of idx
    re = ((x.unsqueeze(2) - near) ** 2).sum(-1)
    assert torch.allclose(d, re, rtol=1e-5, atol=1e-7)                      # dists are the squared distances
    full = torch.cdist(x.double(), y.double()) ** 2
    top = full.topk(K + 1, dim=-1, largest=False)
    clear = (top.values[..., K] - top.values[..., K - 1]) > 1e-6 * top.values[..., K]   # K-th / (K+1)-th not tied
    want = top.indices[..., :K].sort(-1).values
    assert bool((idx.sort(-1).values == want)[clear].all())


def test_chamfer_ten_thousand_pairs_properties(cuda):
    """Config 3 batch shape: directed terms swap under argument swap, translation invariance, identity = 0."""
    P = 10000
    a = _clouds(P, 2048, 5, cuda, -0.5, 0.5)
    b = _clouds(P, 2048, 6, cuda, -0.5, 0.5)
    cd_p, cd_t, f1 = calc_cd(a, b, calc_f1=True)
    cd_p2, cd_t2, f12 = calc_cd(b, a, calc_f1=True)
    assert torch.allclose(cd_t, cd_t2, rtol=1e-6) and torch.allclose(cd_p, cd_p2, rtol=1e-6)
    assert torch.allclose(f1, f12, rtol=1e-6, atol=1e-7)
    z_p, z_t = calc_cd(a[:64], a[:64].clone())
    assert float(z_t.abs().max()) == 0.0 and float(z_p.abs().max()) == 0.0
    # a slice against float64 brute force
    d = torch.cdist(b[:8].double(), a[:8].double()) ** 2                     # chamfer_distance(gt=b, output=a)
    ref_t = d.min(2).values.mean(1) + d.min(1).values.mean(1)
    assert torch.allclose(cd_t[:8].double(), ref_t, rtol=1e-5)


def test_emd_full_size_is_a_transport_plan(cuda):
    B, n = 16, 2048
    a = _clouds(B, n, 7, cuda, -0.5, 0.5)
    b = _clouds(B, n, 8, cuda, -0.5, 0.5)
    cost, match = emd.earth_mover_distance(a, b, return_match=True)          # match (B, m, n)
    assert bool((match >= 0).all())
    # n == m: every point ships (and receives) one unit of mass up to the approximation's residual
    assert float((match.sum(1) - 1).abs().max()) < 2e-2 and float((match.sum(2) - 1).abs().max()) < 2e-2
    d2 = torch.cdist(b.double(), a.double()) ** 2
    ref = (match.double() * d2).sum((1, 2)) / n
    assert torch.allclose(cost.double(), ref, rtol=1e-4)
    fused = emd.earth_mover_distance(a, b)                                    # cost-only path, no matrix
    assert torch.allclose(fused, cost, rtol=1e-4)
    # permuting the points of either cloud does not change the cost beyond the summation-order noise
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(9)).to(cuda)
    assert torch.allclose(emd.earth_mover_distance(a[:, perm].contiguous(), b), fused, rtol=5e-3)


@pytest.mark.timeout(900)
def test_emd_ten_thousand_pair_batch_against_the_oracle(cuda):
    """Config 3 at its real batch shape: ONE call over 10,000 (2048, 3) pairs -- the kernels stride
    pairs over the grid (emd_kernel.cu:41,174-196: `for i = blockIdx.x; i < b; i += gridDim.x`), so pairs far into the
    batch take a code path a 16-pair batch never reaches --, 32 pairs spread over the batch (first / middle / last)
    against the C oracle at north_star's 1e-4; the same pairs in a batch of their own give the same costs."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pdr_oracle as O
    P, n = 10000, 2048
    a = _clouds(P, n, 21, cuda, -0.5, 0.5)
    b = _clouds(P, n, 22, cuda, -0.5, 0.5)
    cost = emd.earth_mover_distance(a, b)
    assert cost.shape == (P,) and bool(torch.isfinite(cost).all())
    pick = sorted(set(list(range(0, 11)) + list(range(4995, 5005)) + list(range(P - 11, P))))[:32]
    an, bn = a[pick].cpu().numpy(), b[pick].cpu().numpy()
    with ThreadPoolExecutor(16) as ex:
        want = np.array(list(ex.map(lambda i: float(O.emd(an[i:i + 1], bn[i:i + 1])[0]), range(len(pick)))))
    np.testing.assert_allclose(cost[pick].cpu().numpy(), want, rtol=1e-4)
    small = emd.earth_mover_distance(a[pick].contiguous(), b[pick].contiguous())
    assert torch.allclose(small, cost[pick], rtol=1e-6)
