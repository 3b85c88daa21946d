"""A reduced PDR network configuration (same structure as the shipped DDPM config,
4 levels, attention everywhere, global + local condition, class embedding) small
enough that its weights and I/O fit in committed fixtures."""
import copy


def tiny_pointnet_config(include_t=True, point_upsample_factor=1):
    cfg = {
        "model_name": "tiny", "in_fea_dim": 0, "partial_in_fea_dim": 1, "out_dim": 3, "include_t": include_t,
        "t_dim": 32, "model.use_xyz": True, "attach_position_to_input_feature": True,
        "include_abs_coordinate": True, "include_center_coordinate": True, "record_neighbor_stats": False,
        "bn_first": False, "bias": True, "res_connect": True, "include_class_condition": True, "num_class": 16,
        "class_condition_dim": 32, "bn": True, "include_local_feature": True, "include_global_feature": True,
        "global_feature_remove_last_activation": False,
        "pnet_global_feature_architecture": [[4, 32, 64], [128, 128]],
        "attention_setting": {"use_attention_module": True, "attention_bn": True,
                              "transform_grouped_feat_out": True, "last_activation": True,
                              "add_attention_to_FeatureMapper_module": True},
        "architecture": {"npoint": [64, 32, 16, 8], "radius": [0.3, 0.5, 0.8, 1.2],
                         "neighbor_definition": "radius", "nsample": [8, 8, 8, 8],
                         "feature_dim": [32, 32, 32, 32, 64], "mlp_depth": 3,
                         "decoder_feature_dim": [32, 32, 32, 32, 64], "include_grouper": False,
                         "decoder_mlp_depth": 2, "use_knn_FP": True, "K": 4},
        "condition_net_architecture": {"npoint": [64, 32, 16, 8], "radius": [0.3, 0.5, 0.8, 1.2],
                                       "neighbor_definition": "radius", "nsample": [8, 8, 8, 8],
                                       "feature_dim": [32, 32, 32, 32, 32], "mlp_depth": 3,
                                       "decoder_feature_dim": [32, 32, 32, 32, 32], "include_grouper": False,
                                       "decoder_mlp_depth": 2, "use_knn_FP": True, "K": 4},
        "feature_mapper_architecture": {"neighbor_definition": "radius",
                                        "encoder_feature_map_dim": [32, 32, 32, 32], "encoder_mlp_depth": 2,
                                        "encoder_radius": [0.3, 0.5, 0.8, 1.2], "encoder_nsample": [8, 8, 8, 8],
                                        "decoder_feature_map_dim": [32, 32, 32, 32, 32], "decoder_mlp_depth": 2,
                                        "decoder_radius": [0.3, 0.5, 0.8, 1.2, 1.6],
                                        "decoder_nsample": [8, 8, 8, 8, 8]},
    }
    if point_upsample_factor > 1:
        cfg["point_upsample_factor"] = point_upsample_factor
        cfg["include_displacement_center_to_final_output"] = False
    return copy.deepcopy(cfg)


def small_fused_config(include_t=True):
    """Like tiny_pointnet_config but every level keeps >= 32 points, the granularity the fused
    channel-last kernels tile over (the shipped configs' smallest level is 64 x K=1)."""
    cfg = tiny_pointnet_config(include_t=include_t)
    for key in ("architecture", "condition_net_architecture"):
        cfg[key]["npoint"] = [128, 64, 32, 32]
    return cfg
