"""Point modules of the dual-path PointNet++ (reference
pointnet2_ops/pointnet2_modules.py): shared 1x1-conv MLP with time / condition
injections, set abstraction (SA), feature propagation (FP, kNN-FP) and the
condition->x_t feature transfer ("FeatureMap") module.

Constructor arguments, forward signatures and parameter names follow the reference
(`first_mlp.0.weight`, `first_mlp.1.group_norm.weight`, `fc`, `fc_condition`,
`fc_second_condition`, `res_connect`, `mlps`, `groupers`, `attention_modules`,
`mapper`, `mlp1`, `mlp2` ...) so a reference checkpoint's state_dict loads
unchanged; the bodies are this project's own and run on the HIP ops.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils
from .attention import AttentionModule, MyGroupNorm


def swish(x):
    return x * torch.sigmoid(x)


class Swish(nn.Module):
    def forward(self, x):
        return swish(x)


def _activation(name):
    if name == 'relu':
        return nn.ReLU(True)
    if name == 'swish':
        return Swish()
    raise AssertionError('activation must be relu or swish')


def build_shared_mlp(mlp_spec, bn=True, bn_first=False, bias=False, activation='relu'):
    """1x1 Conv2d stack.  bn_first=False: [conv, norm, act]*; bn_first=True: [norm, act, conv]*.
    'bn' is GroupNorm(32) on all but the trailing C % 32 channels (pointnet2_modules.py:42-67)."""
    _activation(activation)
    layers = []
    for cin, cout in zip(mlp_spec[:-1], mlp_spec[1:]):
        conv = nn.Conv2d(cin, cout, kernel_size=1, bias=bias)
        if bn_first:
            if bn:
                layers.append(MyGroupNorm(min(32, cin), cin))
            layers += [_activation(activation), conv]
        else:
            layers.append(conv)
            if bn:
                layers.append(MyGroupNorm(32, cout))
            layers.append(_activation(activation))
    return nn.Sequential(*layers)


def _xyz_channels(use_xyz, include_abs, include_center):
    return (3 + (3 if include_abs else 0) + (3 if include_center else 0)) if use_xyz else 0


class Mlp_plus_t_emb(nn.Module):
    """first_mlp -> (+fc(t_emb)) -> second_mlp -> (+fc_condition(c)) -> rest_mlp ->
    (+fc_second_condition(c2)) -> (+residual), all broadcast over (npoint, K)
    (pointnet2_modules.py:69-174)."""

    def __init__(self, mlp_spec, bn, t_dim=128, include_t=True, bn_first=False, bias=False, first_conv=False,
                 first_conv_in_channel=0, res_connect=False, include_condition=False, condition_dim=128,
                 include_second_condition=False, second_condition_dim=128, activation='relu'):
        super().__init__()
        assert len(mlp_spec) >= 3
        if include_second_condition:
            assert len(mlp_spec) >= 4
        self.include_t = include_t
        self.include_condition = include_condition
        self.include_second_condition = include_second_condition
        self.first_conv_bool = first_conv
        self.res_connect_bool = res_connect
        if include_t:
            self.fc = nn.Linear(t_dim, mlp_spec[1])
        if include_condition:
            self.fc_condition = nn.Linear(condition_dim, mlp_spec[2])
        if include_second_condition:
            self.fc_second_condition = nn.Linear(second_condition_dim, mlp_spec[-1])
        if first_conv:
            self.first_conv = nn.Conv2d(first_conv_in_channel, mlp_spec[0], kernel_size=1, bias=bias)
        if res_connect:
            self.res_connect = (None if mlp_spec[0] == mlp_spec[-1]
                                else nn.Conv2d(mlp_spec[0], mlp_spec[-1], kernel_size=1, bias=bias))
        kw = dict(bn_first=bn_first, bias=bias, activation=activation)
        self.first_mlp = build_shared_mlp(mlp_spec[0:2], bn, **kw)
        self.second_mlp = build_shared_mlp(mlp_spec[1:3], bn, **kw)
        self.rest_mlp = build_shared_mlp(mlp_spec[2:], bn, **kw) if len(mlp_spec) > 3 else None

    @staticmethod
    def _inject(h, fc, emb):
        return h + fc(emb).unsqueeze(2).unsqueeze(3)

    def forward(self, feature, t_emb=None, condition_emb=None, second_condition_emb=None):
        if self.first_conv_bool:
            feature = self.first_conv(feature)
        h = self.first_mlp(feature)
        if self.include_t:
            if t_emb is None:
                raise Exception('Should pass t_emb to the forward function')
            h = self._inject(h, self.fc, t_emb)
        elif t_emb is not None:
            raise Exception('This module does not include t but t_emb is given')
        h = self.second_mlp(h)
        if self.include_condition:
            if condition_emb is None:
                raise Exception('Should pass condition_emb to the forward function')
            h = self._inject(h, self.fc_condition, condition_emb)
        elif condition_emb is not None:
            raise Exception('This module does not include condition but condition_emb is given')
        if self.rest_mlp is not None:
            h = self.rest_mlp(h)
        if self.include_second_condition:
            if second_condition_emb is None:
                raise Exception('Should pass second_condition_emb to the forward function')
            h = self._inject(h, self.fc_second_condition, second_condition_emb)
        elif second_condition_emb is not None:
            raise Exception('This module does not include condition but condition_emb is given')
        if self.res_connect_bool:
            h = h + (feature if self.res_connect is None else self.res_connect(feature))
        return h


def pooling_features(feature, count=None, pooling='max'):
    """(B,C,npoint,K) -> (B,C,npoint) by max / masked mean / half-and-half."""
    assert pooling in ['max', 'avg', 'avg_max', 'max_avg']
    K = feature.size(3)
    if pooling == 'max':
        return F.max_pool2d(feature, kernel_size=[1, K]).squeeze(-1)
    if pooling == 'avg':
        return pointnet2_utils.average_feature(feature, count, K)
    half = int(feature.shape[1] / 2)
    mx = F.max_pool2d(feature[:, :half], kernel_size=[1, K]).squeeze(-1)
    av = pointnet2_utils.average_feature(feature[:, half:], count, K)
    return torch.cat([mx, av], dim=1)


def _attention_from(setting, c_query, c_key, c_out):
    return AttentionModule(c_query, c_key, c_query, c_key, c_out, attention_bn=setting['attention_bn'],
                           transform_grouped_feat_out=setting['transform_grouped_feat_out'],
                           last_activation=setting['last_activation'])


def _no_global_attention(setting):
    if setting is not None and setting.get('use_global_attention_module', False):
        raise NotImplementedError('GlobalAttentionModule is outside the built hot path (no shipped config uses it)')


class _PointnetSAModuleBase(nn.Module):
    """Set abstraction: FPS -> gather centres -> group -> MLP(+t,+cond) -> attention/pool."""

    def forward(self, xyz, features, t_emb=None, condition_emb=None, second_condition_emb=None, subset=True,
                record_neighbor_stats=False, pooling='max'):
        assert self.npoint is not None
        sel = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), sel).transpose(1, 2).contiguous()
        if self.use_attention_module:
            centre_feat = pointnet2_utils.gather_operation(features, sel)
        t_emb = t_emb if self.include_t else None
        condition_emb = condition_emb if self.include_condition else None
        second_condition_emb = second_condition_emb if self.include_second_condition else None
        outs = []
        for i, grouper in enumerate(self.groupers):
            grouped, count = grouper(xyz, new_xyz, features, subset=subset,
                                     record_neighbor_stats=record_neighbor_stats, return_counts=True)
            h = self.mlps[i](grouped, t_emb=t_emb, condition_emb=condition_emb,
                             second_condition_emb=second_condition_emb)
            if self.use_attention_module:
                outs.append(self.attention_modules[i](centre_feat, grouped, h, count))
            else:
                outs.append(pooling_features(h, count=count, pooling=pooling))
        return new_xyz, torch.cat(outs, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    def __init__(self, npoint, radii, nsamples, mlps, bn=True, use_xyz=True, t_dim=128, include_t=False,
                 include_abs_coordinate=False, include_center_coordinate=False, bn_first=False, bias=False,
                 first_conv=False, first_conv_in_channel=0, res_connect=False, include_condition=False,
                 condition_dim=128, include_second_condition=False, second_condition_dim=128,
                 neighbor_def='radius', activation='relu', attention_setting=None, global_attention_setting=None):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        _no_global_attention(global_attention_setting)
        self.npoint = npoint
        self.include_t, self.t_dim = include_t, t_dim
        self.include_condition, self.condition_dim = include_condition, condition_dim
        self.include_second_condition, self.second_condition_dim = include_second_condition, second_condition_dim
        self.use_attention_module = bool(attention_setting and attention_setting['use_attention_module'])
        self.use_global_attention_module = False
        self.groupers, self.mlps = nn.ModuleList(), nn.ModuleList()
        self.attention_modules = nn.ModuleList() if self.use_attention_module else None
        self.global_attention_modules = None
        extra = _xyz_channels(use_xyz, include_abs_coordinate, include_center_coordinate)
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(
                pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz,
                                              include_abs_coordinate=include_abs_coordinate,
                                              include_center_coordinate=include_center_coordinate,
                                              neighbor_def=neighbor_def)
                if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            spec = list(spec)
            c_query = first_conv_in_channel if first_conv else spec[0]   # before the xyz channels
            conv_in = first_conv_in_channel + extra if first_conv else first_conv_in_channel
            if not first_conv:
                spec[0] += extra
            self.mlps.append(Mlp_plus_t_emb(spec, bn, t_dim=t_dim, include_t=include_t, bn_first=bn_first,
                                            bias=bias, first_conv=first_conv, first_conv_in_channel=conv_in,
                                            res_connect=res_connect, include_condition=include_condition,
                                            condition_dim=condition_dim,
                                            include_second_condition=include_second_condition,
                                            second_condition_dim=second_condition_dim, activation=activation))
            if self.use_attention_module:
                c_key = conv_in if first_conv else spec[0]
                self.attention_modules.append(_attention_from(attention_setting, c_query, c_key, spec[-1]))


class PointnetSAModule(PointnetSAModuleMSG):
    def __init__(self, mlp, npoint=None, radius=None, nsample=None, **kw):
        super().__init__(npoint=npoint, radii=[radius], nsamples=[nsample], mlps=[mlp], **kw)


class PointnetFPModule(nn.Module):
    """Feature propagation by 3-NN inverse-distance interpolation (pointnet2_modules.py:445-576).
    Not used by the shipped configs (all set use_knn_FP) but part of the op surface."""

    def __init__(self, mlp, bn=True, t_dim=128, include_t=False, bn_first=False, bias=False, first_conv=False,
                 first_conv_in_channel=0, res_connect=False, include_condition=False, condition_dim=128,
                 include_second_condition=False, second_condition_dim=128, include_grouper=False, radius=0,
                 nsample=32, use_xyz=True, include_abs_coordinate=True, include_center_coordinate=False,
                 neighbor_def='radius', activation='relu'):
        super().__init__()
        self.include_t, self.t_dim = include_t, t_dim
        self.include_condition, self.condition_dim = include_condition, condition_dim
        self.include_second_condition, self.second_condition_dim = include_second_condition, second_condition_dim
        self.include_grouper = include_grouper
        mlp = list(mlp)
        if include_grouper:
            extra = _xyz_channels(use_xyz, include_abs_coordinate, include_center_coordinate)
            if first_conv:
                first_conv_in_channel += extra
            else:
                mlp[0] += extra
            self.grouper = pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz,
                                                         include_abs_coordinate=include_abs_coordinate,
                                                         include_center_coordinate=include_center_coordinate,
                                                         neighbor_def=neighbor_def)
        self.mlp = Mlp_plus_t_emb(mlp, bn, t_dim=t_dim, include_t=include_t, bn_first=bn_first, bias=bias,
                                  first_conv=first_conv, first_conv_in_channel=first_conv_in_channel,
                                  res_connect=res_connect, include_condition=include_condition,
                                  condition_dim=condition_dim, include_second_condition=include_second_condition,
                                  second_condition_dim=second_condition_dim, activation=activation)

    def forward(self, unknown, known, unknow_feats, known_feats, t_emb=None, condition_emb=None,
                second_condition_emb=None, record_neighbor_stats=False, pooling='max'):
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            recip = 1.0 / (dist + 1e-8)
            weight = recip / torch.sum(recip, dim=2, keepdim=True)
            interpolated = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated = known_feats.expand(*(list(known_feats.size()[0:2]) + [unknown.size(1)]))
        feats = interpolated if unknow_feats is None else torch.cat([interpolated, unknow_feats], dim=1)
        if self.include_grouper:
            feats, count = self.grouper(unknown, unknown, feats, subset=True,
                                        record_neighbor_stats=record_neighbor_stats, return_counts=True)
        else:
            feats = feats.unsqueeze(-1)
        feats = self.mlp(feats, t_emb=t_emb if self.include_t else None,
                         condition_emb=condition_emb if self.include_condition else None,
                         second_condition_emb=second_condition_emb if self.include_second_condition else None)
        if self.include_grouper:
            return pooling_features(feats, count=count, pooling=pooling)
        return feats.squeeze(-1)


class FeatureMapModule(nn.Module):
    """Feature transfer: ball-query the condition cloud's features (at `xyz`) around the x_t
    points `new_xyz`, MLP, attention-pool with the x_t features as query
    (pointnet2_modules.py:579-649)."""

    def __init__(self, mlp, radius, K, use_xyz=True, include_abs_coordinate=True, include_center_coordinate=False,
                 bn=True, bn_first=True, bias=True, res_connect=True, first_conv=False, first_conv_in_channel=0,
                 neighbor_def='radius', activation='relu', attention_setting=None, query_feature_dim=None):
        super().__init__()
        self.use_attention_module = bool(attention_setting and attention_setting['use_attention_module'])
        mlp = list(mlp)
        extra = _xyz_channels(use_xyz, include_abs_coordinate, include_center_coordinate)
        if first_conv:
            first_conv_in_channel += extra
        else:
            mlp[0] += extra
        self.mlp = Mlp_plus_t_emb(mlp, bn, include_t=False, bn_first=bn_first, bias=bias, first_conv=first_conv,
                                  first_conv_in_channel=first_conv_in_channel, res_connect=res_connect,
                                  include_condition=False, activation=activation)
        self.mapper = pointnet2_utils.QueryAndGroup(radius, K, use_xyz=use_xyz,
                                                    include_abs_coordinate=include_abs_coordinate,
                                                    include_center_coordinate=include_center_coordinate,
                                                    neighbor_def=neighbor_def)
        if self.use_attention_module:
            c_key = first_conv_in_channel if first_conv else mlp[0]
            self.attention_module = _attention_from(attention_setting, query_feature_dim, c_key, mlp[-1])

    def forward(self, xyz, features, new_xyz, subset=False, record_neighbor_stats=True, pooling='max',
                features_at_new_xyz=None):
        grouped, count = self.mapper(xyz, new_xyz, features, subset=subset,
                                     record_neighbor_stats=record_neighbor_stats, return_counts=True)
        h = self.mlp(grouped)
        if self.use_attention_module:
            return self.attention_module(features_at_new_xyz, grouped, h, count)
        return pooling_features(h, count=count, pooling=pooling)


class PointnetKnnFPModule(nn.Module):
    """Feature propagation through K nearest neighbours: group_knn (+11 geometric channels) ->
    mlp1 -> attention-pool (skip features as query) -> cat skip, xyz -> mlp2(+t,+cond)
    (pointnet2_modules.py:652-839)."""

    def __init__(self, mlp1, mlp2, K, bn=True, t_dim=128, include_t=False, bn_first=False, bias=False,
                 first_conv=False, first_conv_in_channel1=0, first_conv_in_channel2=0, res_connect=False,
                 include_condition=False, condition_dim=128, include_second_condition=False,
                 second_condition_dim=128, include_grouper=False, radius=0, nsample=32, use_xyz=True,
                 include_abs_coordinate=True, include_center_coordinate=False, neighbor_def='radius',
                 activation='relu', attention_setting=None, global_attention_setting=None):
        super().__init__()
        _no_global_attention(global_attention_setting)
        self.include_t, self.t_dim = include_t, t_dim
        self.include_condition, self.condition_dim = include_condition, condition_dim
        self.include_second_condition, self.second_condition_dim = include_second_condition, second_condition_dim
        self.K = K
        self.include_grouper = include_grouper
        self.use_global_attention_module = False
        mlp1, mlp2 = list(mlp1), list(mlp2)
        if first_conv:
            first_conv_in_channel1 += 11
        else:
            mlp1[0] += 11
        # mlp1 takes the class embedding through its `fc_condition` slot
        self.mlp1 = Mlp_plus_t_emb(mlp1, bn, t_dim=t_dim, include_t=False, bn_first=bn_first, bias=bias,
                                   first_conv=first_conv, first_conv_in_channel=first_conv_in_channel1,
                                   res_connect=res_connect, include_condition=include_second_condition,
                                   condition_dim=second_condition_dim, activation=activation)
        self.use_attention_module = bool(attention_setting and attention_setting['use_attention_module'])
        if self.use_attention_module:
            c_query = (first_conv_in_channel2 if first_conv else mlp2[0]) - mlp1[-1]   # width of the skip features
            c_key = first_conv_in_channel1 if first_conv else mlp1[0]
            self.attention_module = _attention_from(attention_setting, c_query, c_key, mlp1[-1])
        if include_grouper:
            extra = _xyz_channels(use_xyz, include_abs_coordinate, include_center_coordinate)
            self.grouper = pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz,
                                                         include_abs_coordinate=include_abs_coordinate,
                                                         include_center_coordinate=include_center_coordinate,
                                                         neighbor_def=neighbor_def)
        else:
            extra = 3
        if first_conv:
            first_conv_in_channel2 += extra
        else:
            mlp2[0] += extra
        self.mlp2 = Mlp_plus_t_emb(mlp2, bn, t_dim=t_dim, include_t=include_t, bn_first=bn_first, bias=bias,
                                   first_conv=first_conv, first_conv_in_channel=first_conv_in_channel2,
                                   res_connect=res_connect, include_condition=include_condition,
                                   condition_dim=condition_dim, activation=activation)

    def forward(self, unknown, known, unknow_feats, known_feats, t_emb=None, condition_emb=None,
                second_condition_emb=None, record_neighbor_stats=False, pooling='max'):
        if self.use_attention_module:
            assert known is not None and unknown is not None
        if known is not None:
            grouped = pointnet2_utils.group_knn(unknown, known, known_feats, self.K, transpose=True)
            c2 = second_condition_emb if self.include_second_condition else None
            h = self.mlp1(grouped, t_emb=None, condition_emb=c2)
            if self.use_attention_module:
                interpolated = self.attention_module(unknow_feats, grouped, h, count='all')
            else:
                interpolated = pooling_features(h, count='all', pooling=pooling)
        else:
            interpolated = known_feats.expand(*(list(known_feats.size()[0:2]) + [unknown.size(1)]))
        feats = interpolated if unknow_feats is None else torch.cat([interpolated, unknow_feats], dim=1)
        if self.include_grouper:
            feats, count = self.grouper(unknown, unknown, feats, subset=True,
                                        record_neighbor_stats=record_neighbor_stats, return_counts=True)
        else:
            feats = torch.cat([feats, unknown.transpose(1, 2)], dim=1).unsqueeze(-1)
        feats = self.mlp2(feats, t_emb=t_emb if self.include_t else None,
                          condition_emb=condition_emb if self.include_condition else None)
        if self.include_grouper:
            return pooling_features(feats, count=count, pooling=pooling)
        return feats.squeeze(-1)
