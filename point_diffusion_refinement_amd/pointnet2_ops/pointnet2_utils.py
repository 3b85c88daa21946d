"""Python surface of the point ops -- same names, signatures and return tuples as
the reference's pointnet2_ops/pointnet2_utils.py, running on libpdr_hip.so.

    furthest_point_sample, gather_operation, three_nn, three_interpolate,
    grouping_operation, ball_query           (pointnet2_utils.py:93,129,164,219,268,304)
    QueryAndGroup, GroupAll, group_knn       (:307,441,487)
    count_to_mask, average_feature           (:36,46)

kNN comes from this package's own `_ext.knn_points` (the reference imports
pytorch3d.ops.knn, :7).  Autograd: FPS / ball_query / three_nn are
non-differentiable as in the reference; gather / group / three_interpolate have
backward kernels.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import _ext


def count_to_mask(count, K):
    """(B,npoint) counts -> bool mask (B,npoint,K): slot k valid iff k < count."""
    ar = torch.arange(K, device=count.device, dtype=count.dtype)
    return ar.view(1, 1, K) < count.unsqueeze(-1)


def average_feature(feature, count, K):
    """Masked mean over the neighbour axis. feature (B,C,npoint,K); count (B,npoint) or 'all'."""
    if isinstance(count, str) and count == 'all':
        return F.avg_pool2d(feature, kernel_size=[1, feature.size(3)]).squeeze(-1)
    count = torch.clamp(count, min=1)
    mask = count_to_mask(count, K).unsqueeze(1)
    return (feature * mask).sum(dim=-1) / count.unsqueeze(1)


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        out = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return ()


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n = features.size(2)
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.gather_points_grad(grad_out.contiguous(), idx, ctx.n), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        dist2, idx = _ext.three_nn(unknown, known)
        dist = torch.sqrt(dist2)  # reference returns the L2 distance, not its square (:152-153)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, grad_dist, grad_idx):
        return ()


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.save_for_backward(idx, weight)
        ctx.m = features.size(2)
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        g = _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m)
        return g, torch.zeros_like(idx), torch.zeros_like(weight)


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n = features.size(2)
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.group_points_grad(grad_out.contiguous(), idx, ctx.n), torch.zeros_like(idx)


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        # NOTE the native op takes the queries first (bindings: ball_query(new_xyz, xyz, r, ns))
        idx, counts = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(idx, counts)
        return idx, counts

    @staticmethod
    def backward(ctx, *grads):
        return ()


ball_query = BallQuery.apply


def knn_gather(x, idx):
    """x (B,M,C), idx (B,N,K) int64 -> (B,N,K,C) = x[b, idx[b,n,k], :]."""
    B, M, C = x.shape
    _, N, K = idx.shape
    flat = idx.reshape(B, N * K, 1).expand(-1, -1, C)
    return x.gather(1, flat).view(B, N, K, C)


class QueryAndGroup(nn.Module):
    """Neighbourhood grouping around `new_xyz` (radius ball or K nearest).

    Output channels: [grouped features | relative xyz | (absolute xyz) | (centre xyz)].
    With subset=False a query without any neighbour is given itself as its only
    neighbour with an all-zero feature (reference :376-386, 404-410).
    """

    def __init__(self, radius, nsample, use_xyz=True, include_abs_coordinate=False,
                 include_center_coordinate=False, neighbor_def='radius'):
        super().__init__()
        if neighbor_def not in ('radius', 'nn'):
            raise AssertionError('neighbor_def must be radius or nn')
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.include_abs_coordinate = include_abs_coordinate
        self.include_center_coordinate = include_center_coordinate
        self.neighbor_def = neighbor_def
        self.neighbor_stats = None
        self.neighbor_num_quantile = None
        self.quantile = torch.linspace(0, 1, 11)

    def _neighbours(self, xyz, new_xyz):
        if self.neighbor_def == 'radius':
            return ball_query(self.radius, self.nsample, xyz, new_xyz)
        k = min(self.nsample, xyz.shape[1])
        _, idx, _ = _ext.knn_points(new_xyz, xyz, k)
        idx = idx.int()
        counts = torch.ones(idx.shape[0], idx.shape[1], device=new_xyz.device) * k
        return idx, counts

    def forward(self, xyz, new_xyz, features=None, subset=True, record_neighbor_stats=False,
                return_counts=False):
        idx, counts = self._neighbours(xyz, new_xyz)
        radius_mode = self.neighbor_def == 'radius'
        centre = new_xyz.transpose(1, 2).unsqueeze(-1)                     # (B,3,np,1)
        abs_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)  # (B,3,np,K)
        patch_empty = (not subset) and radius_mode
        if patch_empty:
            have = (counts > 0).float().unsqueeze(1).unsqueeze(-1).detach()
            none = 1 - have
            abs_xyz = have * abs_xyz + none * centre
        parts = [abs_xyz - centre]
        if self.include_abs_coordinate:
            parts.append(abs_xyz)
        if self.include_center_coordinate:
            parts.append(centre.expand(-1, -1, -1, abs_xyz.shape[3]))
        grouped_xyz = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)

        if features is not None:
            grouped = grouping_operation(features, idx)
            if patch_empty:
                zero = torch.zeros(features.shape[1], device=features.device).view(1, -1, 1, 1)
                grouped = have * grouped + none * zero
            new_features = torch.cat([grouped, grouped_xyz], dim=1) if self.use_xyz else grouped
        else:
            if not self.use_xyz:
                raise AssertionError("Cannot have not features and not use xyz as a feature!")
            new_features = grouped_xyz

        if record_neighbor_stats:
            with torch.no_grad():
                c = counts.float()
                self.neighbor_stats = torch.stack([c.min(), c.mean(), c.max()])
                self.neighbor_num_quantile = torch.quantile(c, self.quantile.to(c.device)).long()

        if return_counts:
            return new_features, ('all' if not radius_mode else counts)
        return new_features


class GroupAll(nn.Module):
    """Single group holding the whole cloud: (B, C(+3), 1, N)."""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        gxyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return gxyz
        gfeat = features.unsqueeze(2)
        return torch.cat([gfeat, gxyz], dim=1) if self.use_xyz else gfeat


def group_knn(x, y, features_at_y, K, transpose=False):
    """K nearest neighbours of x (B,N1,3) in y (B,N2,3) with 11 geometric channels.

    Returns (B,N1,K,C+11) = [feats | d2 | normalised 1/(d2+1e-8) | nn_abs(3) | nn_rel(3) | x(3)],
    or its (B,C+11,N1,K) view when transpose=True (features_at_y then is (B,C,N2)).
    The inverse-distance weights use the SQUARED distances (reference :500-503).
    """
    feats_y = features_at_y.transpose(1, 2).contiguous() if transpose else features_at_y
    d2, idx, nn_abs = _ext.knn_points(x, y, K, return_nn=True)
    # K > N2: knn_points pads missing neighbours with idx -1 (dist 0, nn 0).  pytorch3d zero-initialises idx, so
    # the reference's knn_gather reads row 0 there; clamp to reproduce that instead of an out-of-range gather.
    neigh = knn_gather(feats_y, idx.clamp(min=0))
    x_rep = x.unsqueeze(2).repeat(1, 1, K, 1)
    nn_rel = nn_abs - x_rep
    d2 = d2.unsqueeze(3)
    recip = 1.0 / (d2 + 1e-8)
    weight = recip / torch.sum(recip, dim=2, keepdim=True)
    out = torch.cat([neigh, d2, weight, nn_abs, nn_rel, x_rep], dim=3)
    if transpose:
        out = out.transpose(2, 3).transpose(1, 2)
    return out
