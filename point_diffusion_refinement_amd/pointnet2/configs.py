"""Network / diffusion configuration of the shipped DDPM experiment, as Python data.

Values transcribed from the reference's
exp_configs/mvp_configs/config_standard_attention_real_3072_partial_points_rot_90_scale_1.2_translation_0.1.json
(diffusion_config :2-6, pointnet_config :7-78) with the string-encoded lists already
restored (json_reader.py:14-24).  Only the sections the sampling path reads are kept."""
import copy

DIFFUSION_CONFIG = {"T": 1000, "beta_0": 0.0001, "beta_T": 0.02}

_ATTENTION = {"use_attention_module": True, "attention_bn": True, "transform_grouped_feat_out": True,
              "last_activation": True, "add_attention_to_FeatureMapper_module": True}

_POINTNET_CONFIG = {
    "model_name": "shape_completion_mirror_rot_90_scale_1.2_translation_0.1",
    "in_fea_dim": 0, "partial_in_fea_dim": 1, "out_dim": 3, "include_t": True, "t_dim": 128,
    "model.use_xyz": True, "attach_position_to_input_feature": True, "include_abs_coordinate": True,
    "include_center_coordinate": True, "record_neighbor_stats": False, "bn_first": False, "bias": True,
    "res_connect": True, "include_class_condition": True, "num_class": 16, "class_condition_dim": 128,
    "bn": True, "include_local_feature": True, "include_global_feature": True,
    "global_feature_remove_last_activation": False,
    "pnet_global_feature_architecture": [[4, 128, 256], [512, 1024]],
    "attention_setting": _ATTENTION,
    "architecture": {
        "npoint": [1024, 256, 64, 16], "radius": [0.1, 0.2, 0.4, 0.8], "neighbor_definition": "radius",
        "nsample": [32, 32, 32, 32], "feature_dim": [32, 64, 128, 256, 512], "mlp_depth": 3,
        "decoder_feature_dim": [128, 128, 256, 256, 512], "include_grouper": False, "decoder_mlp_depth": 2,
        "use_knn_FP": True, "K": 8},
    "condition_net_architecture": {
        "npoint": [1024, 256, 64, 16], "radius": [0.1, 0.2, 0.4, 0.8], "neighbor_definition": "radius",
        "nsample": [32, 32, 32, 32], "feature_dim": [32, 32, 64, 64, 128], "mlp_depth": 3,
        "decoder_feature_dim": [32, 32, 64, 64, 128], "include_grouper": False, "decoder_mlp_depth": 2,
        "use_knn_FP": True, "K": 8},
    "feature_mapper_architecture": {
        "neighbor_definition": "radius", "encoder_feature_map_dim": [32, 32, 64, 64], "encoder_mlp_depth": 2,
        "encoder_radius": [0.1, 0.2, 0.4, 0.8], "encoder_nsample": [32, 32, 32, 32],
        "decoder_feature_map_dim": [32, 32, 64, 64, 128], "decoder_mlp_depth": 2,
        "decoder_radius": [0.1, 0.2, 0.4, 0.8, 1.6], "decoder_nsample": [32, 32, 32, 32, 32]},
}


def ddpm_pointnet_config():
    """Fresh (deep-copied) pointnet_config of the T=1000 completion DDPM."""
    return copy.deepcopy(_POINTNET_CONFIG)


def refinement_pointnet_config(point_upsample_factor=1):
    """Refinement network: same architecture, no step embedding (train.py:689-699), optional upsampling."""
    cfg = ddpm_pointnet_config()
    cfg["include_t"] = False
    if point_upsample_factor > 1:
        cfg["point_upsample_factor"] = point_upsample_factor
        cfg["include_displacement_center_to_final_output"] = False
    return cfg


def synthetic_batch(B, N=2048, M=3072, seed=0, device=None):
    """Synthetic inputs of the BASELINE shape contract: x_T ~ N(0,1) (B,N,3); condition
    (B,M,4) = xyz ~ U[-1,1]^3 whose second half is the z-mirrored copy of the first, 4th channel the
    +-1 mirror flag (data_utils/mirror_partial.py:21-33); labels ~ U{0..15}.  CPU generator => the same
    values on every machine."""
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, N, 3, generator=g)
    half = torch.rand(B, M // 2, 3, generator=g) * 2 - 1
    mirrored = half * torch.tensor([1.0, 1.0, -1.0])
    cond = torch.cat([torch.cat([half, torch.ones(B, M // 2, 1)], 2),
                      torch.cat([mirrored, -torch.ones(B, M // 2, 1)], 2)], 1).contiguous()
    label = torch.randint(0, 16, (B,), generator=g)
    if device is not None:
        x, cond, label = x.to(device), cond.to(device), label.to(device)
    return x, cond, label


def synthetic_surface_batch(B, N=2048, M=3072, seed=0, device=None):
    """Synthetic inputs shaped like a real completion job (bench.py's `trajectory` leg, the dense-regime tests): x_0 =
    N points on a torus inside [-0.5, 0.5]^3 (major radius 0.33, minor radius 0.08 - 0.16, own random rotation per cloud --
    the scale of the MVP shapes, whose neighbourhood radii the architecture's 0.1 ... 1.6 were chosen for), condition =
    its partial view (the M / 2 points nearest to a random viewpoint direction) mirrored about z with the +-1 flag
    channel (data_utils/mirror_partial.py:21-33), labels ~ U{0..15}.  With `q_sample` below this gives the marginal
    x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) eps a trained network's trajectory follows (reference util.py:280-282)
    without a checkpoint.  CPU generator => the same values on every machine."""
    import math
    import torch
    g = torch.Generator().manual_seed(1000 + seed)
    u = torch.rand(B, N, generator=g) * 2 * math.pi
    v = torch.rand(B, N, generator=g) * 2 * math.pi
    R = 0.33
    r = 0.08 + 0.08 * torch.rand(B, 1, generator=g)
    p = torch.stack([(R + r * torch.cos(v)) * torch.cos(u), (R + r * torch.cos(v)) * torch.sin(u), r * torch.sin(v)], 2)
    # a random rotation per cloud (QR of a Gaussian matrix)
    q, _ = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))
    x0 = torch.bmm(p, q).contiguous()
    view = torch.nn.functional.normalize(torch.randn(B, 1, 3, generator=g), dim=2)
    order = (x0 * view).sum(2).argsort(dim=1, descending=True)[:, :M // 2]
    half = torch.gather(x0, 1, order[:, :, None].expand(-1, -1, 3))
    mirrored = half * torch.tensor([1.0, 1.0, -1.0])
    cond = torch.cat([torch.cat([half, torch.ones(B, M // 2, 1)], 2),
                      torch.cat([mirrored, -torch.ones(B, M // 2, 1)], 2)], 1).contiguous()
    label = torch.randint(0, 16, (B,), generator=g)
    if device is not None:
        x0, cond, label = x0.to(device), cond.to(device), label.to(device)
    return x0, cond, label


def q_sample(x0, t, diffusion_hyperparams, seed=0):
    """x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) eps (reference util.py:280-282, the training marginal), eps from a CPU
    generator keyed by (seed, t)."""
    import torch
    ab = diffusion_hyperparams["Alpha_bar"][int(t)].item()
    g = torch.Generator().manual_seed(7919 * int(seed) + int(t))
    eps = torch.randn(x0.shape, generator=g).to(x0.device)
    return (ab ** 0.5) * x0 + ((1.0 - ab) ** 0.5) * eps
