"""Seeded input builders shared by the fixture generator and the tests (CPU generator =>
identical values in the build container and on the GPU box)."""
import torch


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def layer_clouds():
    """xyz (2,96,3) source cloud, new_xyz (2,48,3) queries (first 24 are cloud points, some far away
    so that subset=False sees empty balls), feats (2,6,96)."""
    g = _gen(1)
    xyz = torch.rand(2, 96, 3, generator=g) * 2 - 1
    new_xyz = torch.rand(2, 48, 3, generator=g) * 2 - 1
    new_xyz[:, :24] = xyz[:, :24]
    new_xyz[:, 40:] += 3.0
    feats = torch.randn(2, 6, 96, generator=g)
    return xyz.contiguous(), new_xyz.contiguous(), feats.contiguous()


def embeddings():
    g = _gen(2)
    return torch.randn(2, 64, generator=g), torch.randn(2, 40, generator=g), torch.randn(2, 24, generator=g)


def network_inputs(B=2, N=128, M=192):
    """x_t (B,N,3) ~ N(0,1); condition (B,M,4): xyz in [-1,1]^3, second half mirrored (z -> -z) with flag -1
    (mirror_partial.py:21-33 shape contract); ts (B,) float; label (B,) long."""
    g = _gen(3)
    x = torch.randn(B, N, 3, generator=g)
    half = torch.rand(B, M // 2, 3, generator=g) * 2 - 1
    mirrored = half * torch.tensor([1.0, 1.0, -1.0])
    cond = torch.cat([torch.cat([half, torch.ones(B, M // 2, 1)], 2),
                      torch.cat([mirrored, -torch.ones(B, M // 2, 1)], 2)], 1)
    ts = torch.tensor([7.0, 3.0])[:B]
    label = torch.tensor([3, 11])[:B]
    return x.contiguous(), cond.contiguous(), ts, label


def metric_clouds():
    g = _gen(4)
    gt = torch.rand(3, 256, 3, generator=g) - 0.5
    gen = gt[:, torch.randperm(256, generator=g)] + 0.02 * torch.randn(3, 256, 3, generator=g)
    return gen.contiguous(), gt.contiguous()


def ddpm_inputs(B=1, N=2048, M=3072):
    """Full-size single cloud of the BASELINE shape contract (configs[0]): x_t (1,2048,3) ~ N(0,1), condition
    (1,3072,4) mirrored halves with the +-1 flag, ts = 500, label = 5."""
    g = _gen(5)
    x = torch.randn(B, N, 3, generator=g)
    half = torch.rand(B, M // 2, 3, generator=g) * 2 - 1
    mirrored = half * torch.tensor([1.0, 1.0, -1.0])
    cond = torch.cat([torch.cat([half, torch.ones(B, M // 2, 1)], 2),
                      torch.cat([mirrored, -torch.ones(B, M // 2, 1)], 2)], 1)
    ts = torch.full((B,), 500.0)
    label = torch.full((B,), 5, dtype=torch.long)
    return x.contiguous(), cond.contiguous(), ts, label


def refine_coarse(B=1, N=2048):
    """Coarse completed cloud handed to the refinement network: U[-1,1]^3 (the range of completed MVP shapes)."""
    return (torch.rand(B, N, 3, generator=_gen(41)) * 2 - 1).contiguous()


def dense_inputs(B=2):
    """x_0 (B,2048,3) synthetic surfaces (tori in [-0.5, 0.5]^3), condition (B,3072,4) their mirrored partial views,
    labels: the dense / mixed-ball regime of a reverse process' last steps (configs.synthetic_surface_batch, seed 7)."""
    from point_diffusion_refinement_amd.pointnet2.configs import synthetic_surface_batch
    return synthetic_surface_batch(B, seed=7)


def dense_xt(x0, t):
    """x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) eps of the T = 1000 schedule (reference util.py:280-282), eps seeded by t."""
    from point_diffusion_refinement_amd.pointnet2 import util
    from point_diffusion_refinement_amd.pointnet2.configs import DIFFUSION_CONFIG, q_sample
    return q_sample(x0, t, util.calc_diffusion_hyperparams(**DIFFUSION_CONFIG), seed=7).contiguous()
