"""Approximate earth mover's distance -- same surface as reference pointnet2/emd.py
(EarthMoverDistanceFunction :6-28, earth_mover_distance :31-56, EMD_distance :58-72),
running on libpdr_hip.so instead of the `emd_cuda` extension.

    cost = sum_{k,l} |xyz1_k - xyz2_l|^2 * match[l,k] / max(n, m)

When neither the match matrix nor a gradient is requested, the cost comes from the
fused pdr_emd_cost path that never materialises the (B,m,n) matrix (16.8 MB per
2048^2 pair in the reference, rewritten once per temperature level).
"""
import torch
import torch.nn as nn

from .. import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check(x, name):
    assert x.is_cuda, "Only support cuda currently."
    if x.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (the double instantiation of the reference is not built)" % name)
    if x.dim() != 3 or x.shape[2] != 3:
        raise RuntimeError("%s must be (B, N, 3)" % name)


def approxmatch_forward(xyz1, xyz2):
    """emd_cuda.approxmatch_forward: (B,n,3), (B,m,3) -> match (B,m,n)."""
    _check(xyz1, "xyz1"), _check(xyz2, "xyz2")
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    lib = _lib.load()
    match = torch.empty((B, m, n), dtype=torch.float32, device=xyz1.device)
    temp = torch.empty((lib.pdr_emd_workspace_bytes(B, n, m) // 4,), dtype=torch.float32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        _lib.check(lib.pdr_approxmatch(xyz1.data_ptr(), xyz2.data_ptr(), B, n, m, match.data_ptr(), temp.data_ptr(),
                                       _stream()), "approxmatch_forward")
    return match


def matchcost_forward(xyz1, xyz2, match):
    """emd_cuda.matchcost_forward: -> cost (B), not yet divided by max(n,m)."""
    _check(xyz1, "xyz1"), _check(xyz2, "xyz2")
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    lib = _lib.load()
    cost = torch.empty((B,), dtype=torch.float32, device=xyz1.device)
    temp = torch.empty((lib.pdr_matchcost_workspace_bytes(B, n, m) // 4,), dtype=torch.float32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        _lib.check(lib.pdr_matchcost(xyz1.data_ptr(), xyz2.data_ptr(), match.contiguous().data_ptr(), B, n, m,
                                     cost.data_ptr(), temp.data_ptr(), _stream()), "matchcost_forward")
    return cost


def matchcost_backward(grad_cost, xyz1, xyz2, match):
    """emd_cuda.matchcost_backward: -> [grad1 (B,n,3), grad2 (B,m,3)]."""
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = torch.empty_like(xyz1)
    g2 = torch.empty_like(xyz2)
    with torch.cuda.device(xyz1.device):
        _lib.check(_lib.load().pdr_matchcost_grad(grad_cost.contiguous().data_ptr(), xyz1.data_ptr(),
                                                  xyz2.data_ptr(), match.data_ptr(), B, n, m, g1.data_ptr(),
                                                  g2.data_ptr(), _stream()), "matchcost_backward")
    return [g1, g2]


def emd_cost_fused(xyz1, xyz2):
    """matchcost(approxmatch(xyz1, xyz2)) without the match matrix (no autograd)."""
    _check(xyz1, "xyz1"), _check(xyz2, "xyz2")
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    lib = _lib.load()
    cost = torch.empty((B,), dtype=torch.float32, device=xyz1.device)
    temp = torch.empty((lib.pdr_emd_workspace_bytes(B, n, m) // 4,), dtype=torch.float32, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        _lib.check(lib.pdr_emd_cost(xyz1.data_ptr(), xyz2.data_ptr(), B, n, m, cost.data_ptr(), temp.data_ptr(),
                                    _stream()), "emd_cost")
    return cost


class EarthMoverDistanceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2, return_match=False):
        xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
        assert xyz1.is_cuda and xyz2.is_cuda, "Only support cuda currently."
        denom = max(xyz1.shape[1], xyz2.shape[1])
        needs_grad = any(ctx.needs_input_grad[:2])
        if not return_match and not needs_grad:
            return emd_cost_fused(xyz1, xyz2) / denom
        match = approxmatch_forward(xyz1, xyz2)
        cost = matchcost_forward(xyz1, xyz2, match) / denom
        ctx.save_for_backward(xyz1, xyz2, match)
        return (cost, match) if return_match else cost

    @staticmethod
    def backward(ctx, grad_cost, *unused):
        xyz1, xyz2, match = ctx.saved_tensors
        g1, g2 = matchcost_backward(grad_cost.contiguous(), xyz1, xyz2, match)
        return g1, g2, None


def _prep(xyz1, xyz2, transpose):
    if xyz1.dim() == 2:
        xyz1 = xyz1.unsqueeze(0)
    if xyz2.dim() == 2:
        xyz2 = xyz2.unsqueeze(0)
    if transpose:
        xyz1, xyz2 = xyz1.transpose(1, 2), xyz2.transpose(1, 2)
    return xyz1, xyz2


def earth_mover_distance(xyz1, xyz2, transpose=False, return_match=False):
    """xyz1 (b,n,3), xyz2 (b,m,3) [or (b,3,n) with transpose] -> cost (b) [, match (b,m,n)]."""
    xyz1, xyz2 = _prep(xyz1, xyz2, transpose)
    return EarthMoverDistanceFunction.apply(xyz1, xyz2, bool(return_match))


class EMD_distance(nn.Module):
    def forward(self, xyz1, xyz2, transpose=False, return_match=False):
        return earth_mover_distance(xyz1, xyz2, transpose=transpose, return_match=return_match)
