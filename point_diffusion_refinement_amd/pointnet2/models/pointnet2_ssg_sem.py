"""Shared pieces of the PointNet++ networks: sinusoidal step embedding and the
SA / FP stack builders (reference pointnet2/models/pointnet2_ssg_sem.py:14-31,
42-45, 47-177).  The unconditional PointNet2SemSegSSG.forward is outside the hot
path (every shipped config is conditioned on the partial cloud) and is not built."""
import numpy as np
import torch
import torch.nn as nn

from ...pointnet2_ops.pointnet2_modules import PointnetFPModule, PointnetKnnFPModule, PointnetSAModule


def swish(x):
    return x * torch.sigmoid(x)


_FREQ_CACHE = {}


def _frequencies(half, device):
    # computed on the CPU exactly like the reference (so the f32 values are identical), then kept
    # resident per device: no per-step H2D copy, and the step stays hipGraph-capturable
    key = (half, str(device))
    if key not in _FREQ_CACHE:
        _FREQ_CACHE[key] = torch.exp(torch.arange(half) * -(np.log(10000) / (half - 1))).to(device)
    return _FREQ_CACHE[key]


def calc_t_emb(ts, t_emb_dim):
    """(B,) float steps -> (B, t_emb_dim) [sin(t w_i) | cos(t w_i)], w_i = 10000^(-i/(half-1))."""
    assert t_emb_dim % 2 == 0
    half = t_emb_dim // 2
    freq = _frequencies(half, ts.device)
    arg = ts.unsqueeze(1) * freq
    return torch.cat((torch.sin(arg), torch.cos(arg)), 1)


class PointNet2SemSegSSG(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams
        self._build_model()

    def _build_model(self):
        raise NotImplementedError('only the cloud-conditioned network is built; see PointNet2CloudCondition')

    @staticmethod
    def _break_up_pc(pc):
        xyz = pc[..., 0:3].contiguous()
        feats = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, feats

    def _condition_slots(self, include_class_condition, class_condition_dim, include_global_feature,
                         global_feature_dim):
        """Which embedding feeds fc_condition / fc_second_condition (ssg_sem.py:73-82)."""
        cls_dim = self.hparams["class_condition_dim"] if class_condition_dim is None else class_condition_dim
        if include_global_feature:
            return dict(include_condition=True, condition_dim=global_feature_dim,
                        include_second_condition=include_class_condition, second_condition_dim=cls_dim)
        return dict(include_condition=include_class_condition, condition_dim=cls_dim,
                    include_second_condition=False, second_condition_dim=None)

    @staticmethod
    def _layer_global_attention(setting, i):
        if setting is not None and setting['use_global_attention_module'] and \
                i in setting['global_attention_layer_index']:
            return setting
        return None

    def build_SA_model(self, npoint, radius, nsample, feature_dim, mlp_depth, in_fea_dim, include_t,
                       include_class_condition, class_condition_dim=None, include_global_feature=False,
                       global_feature_dim=None, additional_fea_dim=None, neighbor_def='radius', activation='relu',
                       bn=True, attention_setting=None, global_attention_setting=None):
        hp = self.hparams
        if not isinstance(neighbor_def, list):
            neighbor_def = [neighbor_def] * len(radius)
        slots = self._condition_slots(include_class_condition, class_condition_dim, include_global_feature,
                                      global_feature_dim)
        stack = nn.ModuleList()
        for i in range(len(npoint)):
            spec = [feature_dim[i]] * mlp_depth + [feature_dim[i + 1]]
            if additional_fea_dim is not None:
                spec[0] += additional_fea_dim[i]
            first_conv = hp["bn_first"] and i == 0
            if i == 0 and not first_conv:
                spec[0] = in_fea_dim
            stack.append(PointnetSAModule(
                npoint=npoint[i], radius=radius[i], nsample=nsample[i], mlp=spec, use_xyz=hp["model.use_xyz"],
                t_dim=4 * hp['t_dim'], include_t=include_t, include_abs_coordinate=self.include_abs_coordinate,
                include_center_coordinate=hp.get("include_center_coordinate", False), bn_first=hp["bn_first"],
                first_conv=first_conv, first_conv_in_channel=in_fea_dim, res_connect=hp["res_connect"],
                bias=hp["bias"], neighbor_def=neighbor_def[i], activation=activation, bn=bn,
                attention_setting=attention_setting,
                global_attention_setting=self._layer_global_attention(global_attention_setting, i), **slots))
        return stack

    def build_FP_model(self, decoder_feature_dim, decoder_mlp_depth, feature_dim, in_fea_dim, include_t,
                       include_class_condition, class_condition_dim=None, include_global_feature=False,
                       global_feature_dim=None, additional_fea_dim=None, use_knn_FP=False, K=3,
                       include_grouper=False, radius=[0], nsample=[32], neighbor_def='radius', activation='relu',
                       bn=True, attention_setting=None, global_attention_setting=None):
        hp = self.hparams
        if not isinstance(neighbor_def, list):
            neighbor_def = [neighbor_def] * len(radius)
        slots = self._condition_slots(include_class_condition, class_condition_dim, include_global_feature,
                                      global_feature_dim)
        stack = nn.ModuleList()
        for i in range(len(decoder_feature_dim) - 1):
            skip = in_fea_dim if i == 0 else feature_dim[i]
            common = dict(first_conv=False, bn=bn, t_dim=4 * hp['t_dim'], include_t=include_t,
                          bn_first=hp["bn_first"], res_connect=hp["res_connect"], bias=hp["bias"],
                          include_grouper=include_grouper, radius=radius[i], nsample=nsample[i],
                          use_xyz=hp["model.use_xyz"], include_abs_coordinate=self.include_abs_coordinate,
                          include_center_coordinate=hp.get("include_center_coordinate", False),
                          neighbor_def=neighbor_def[i], activation=activation, **slots)
            if use_knn_FP:
                mlp1 = [decoder_feature_dim[i + 1]] + [decoder_feature_dim[i]] * decoder_mlp_depth
                mlp2 = [decoder_feature_dim[i] + skip] + [decoder_feature_dim[i]] * decoder_mlp_depth
                if additional_fea_dim is not None:
                    mlp1[0] += additional_fea_dim[i]
                stack.append(PointnetKnnFPModule(
                    mlp1=mlp1, mlp2=mlp2, K=K, attention_setting=attention_setting,
                    global_attention_setting=self._layer_global_attention(global_attention_setting, i), **common))
            else:
                mlp = [decoder_feature_dim[i + 1] + skip] + [decoder_feature_dim[i]] * decoder_mlp_depth
                if additional_fea_dim is not None:
                    mlp[0] += additional_fea_dim[i]
                stack.append(PointnetFPModule(mlp=mlp, **common))
        return stack
