"""Training-side data parallelism (reference distributed.py:94-146) on 2 gloo ranks: bucketed, overlapped gradient
averaging == the gradient of the mean loss over the concatenated batch in one process."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from point_diffusion_refinement_amd.pointnet2.distributed import apply_gradient_allreduce


def _model(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.ReLU(), torch.nn.Linear(33, 19), torch.nn.ReLU(),
                               torch.nn.Linear(19, 3))


def _data():
    g = torch.Generator().manual_seed(5)
    return torch.randn(8, 7, generator=g), torch.randn(8, 3, generator=g)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = apply_gradient_allreduce(_model(100 + rank), bucket_bytes=1024)     # different init: broadcast fixes it
        x, y = _data()
        lo, hi = rank * 4, rank * 4 + 4
        for step in range(2):                                                     # hooks re-arm every backward
            net.zero_grad()
            ((net(x[lo:hi]) - y[lo:hi]) ** 2).mean().backward()
        out[rank] = [p.grad.numpy().copy() for p in net.parameters()] + \
                    [p.detach().numpy().copy() for p in net.parameters()]
        assert len(net._pdr_grad_allreduce.buckets) > 1
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_average_equals_full_batch_gradient():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    ref = _model(100)                                                             # rank 0's initial weights
    x, y = _data()
    ((ref(x) - y) ** 2).mean().backward()
    want = [p.grad.numpy() for p in ref.parameters()] + [p.detach().numpy() for p in ref.parameters()]
    for r in range(2):
        for a, b in zip(out[r], want):
            np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-7)


class _Branchy(torch.nn.Module):
    """`used` feeds the loss on every rank, `rank1_only` on rank 1 only, `never` on no rank."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.used = torch.nn.Linear(5, 4)
        self.rank1_only = torch.nn.Linear(5, 4)
        self.never = torch.nn.Linear(5, 4)

    def forward(self, x, extra):
        y = self.used(x)
        return y + self.rank1_only(x) if extra else y


def _worker_unused(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = apply_gradient_allreduce(_Branchy(), bucket_bytes=64)
        x = torch.arange(10.0).reshape(2, 5) / 10 + rank
        net.zero_grad(set_to_none=True)
        net(x, extra=rank == 1).sum().backward()
        out[rank] = {n: (None if p.grad is None else p.grad.numpy().copy()) for n, p in net.named_parameters()}
    finally:
        dist.destroy_process_group()


def test_parameters_no_rank_used_keep_grad_none_and_partially_used_ones_are_averaged():
    """a parameter without a gradient on ANY rank must keep .grad = None (reference distributed.py:112
    skips it; weight decay / momentum must not touch it), one used on SOME ranks gets the same averaged gradient on
    every rank."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = mp.Manager().dict()
    mp.spawn(_worker_unused, args=(2, port, out), nprocs=2, join=True)
    x1 = torch.arange(10.0).reshape(2, 5) / 10 + 1
    for r in range(2):
        assert out[r]["never.weight"] is None and out[r]["never.bias"] is None
        # rank 1's gradient of sum(W x + b) w.r.t. W = column sums of x, averaged over 2 ranks (rank 0 contributes 0)
        np.testing.assert_allclose(out[r]["rank1_only.weight"], np.tile(x1.sum(0).numpy() / 2, (4, 1)), rtol=1e-6)
        np.testing.assert_allclose(out[r]["rank1_only.bias"], np.full(4, 2 / 2.0), rtol=1e-6)
        np.testing.assert_allclose(out[r]["used.weight"], out[0]["used.weight"])
