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
