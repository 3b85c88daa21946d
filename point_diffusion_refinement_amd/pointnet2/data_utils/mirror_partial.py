"""Mirror-and-concat preprocessing of partial clouds -- same surface as reference
pointnet2/data_utils/mirror_partial.py (mirror :5-9, down_sample_points :11-20, mirror_and_concat :22-38),
the producer of the (B, 3072, 4) condition clouds of the DDPM configs
(mvp_dataloader/generate_mirrored_partial.py:44).

A partial cloud is reflected about a coordinate plane, the copy is tagged with a 4th channel (-1; originals +1),
both are concatenated and farthest-point-sampled (on xyz only) down to the requested sizes.  FPS and the row
gather run on libpdr_hip.so; tensors stay on the input's device (the reference moves them with `.cuda()`).
"""
import torch

from ...pointnet2_ops import pointnet2_utils


def mirror(partial, axis=1):
    """(B,N,3) -> copy with coordinate `axis` negated."""
    sign = torch.ones(partial.shape[-1], dtype=partial.dtype, device=partial.device)
    sign[axis] = -1
    return partial * sign


def down_sample_points(xyz, npoints):
    """(B,N,4) -> (B,npoints,4): FPS on the first three channels, all channels of the picked rows."""
    idx = pointnet2_utils.furthest_point_sample(xyz[:, :, 0:3].contiguous(), npoints)
    picked = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), idx)      # (B,4,npoints)
    return picked.transpose(1, 2).contiguous()


def mirror_and_concat(partial, axis=2, num_points=(2048, 3072)):
    """(B,N,3) -> ((B,2N,4) tagged concat, then one (B,n,4) down-sampled cloud per entry of num_points)."""
    B, N, _ = partial.shape
    tag = torch.ones(B, N, 1, dtype=partial.dtype, device=partial.device)
    both = torch.cat([torch.cat([partial, tag], 2), torch.cat([mirror(partial, axis), -tag], 2)], 1).contiguous()
    return (both,) + tuple(down_sample_points(both, n) for n in num_points)
