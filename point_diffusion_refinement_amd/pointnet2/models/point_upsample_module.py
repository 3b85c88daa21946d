"""Displacement -> refined / upsampled cloud (reference models/point_upsample_module.py:4-28)."""
import numpy as np
import torch


def point_upsample(coarse, displacement, point_upsample_factor, include_displacement_center_to_final_output,
                   output_scale_factor_value):
    """coarse (B,N,3); displacement (B,N,3*(f+1)) [or 3*f when the refined centre is itself
    one of the f outputs].  Returns (refined (B,N*f,3), refined centres (B,N,3))."""
    B, N, _ = coarse.size()
    centre = coarse + displacement[:, :, 0:3] * output_scale_factor_value
    per_point = point_upsample_factor - 1 if include_displacement_center_to_final_output else point_upsample_factor
    grid = (displacement[:, :, 3:] * (1 / np.sqrt(point_upsample_factor))).view(B, N, per_point, 3)
    up = (centre.unsqueeze(2) + grid * output_scale_factor_value).reshape(B, -1, 3)
    if include_displacement_center_to_final_output:
        return torch.cat([up, centre], dim=1).contiguous(), centre
    return up.contiguous(), centre
