"""Dual-path conditional PointNet++: the epsilon-network of the DDPM (and, without
the step embedding, the refinement network) of Point Diffusion-Refinement.

Mirrors reference pointnet2/models/pointnet2_with_pcld_condition.py:
    PointNet2CloudCondition(hparams).forward(pointcloud, condition, ts, label,
                                             use_retained_condition_feature)   (:276)
    .reset_cond_features()                                                      (:270)
One PointNet++ encodes the mirrored partial cloud (B,3072,4); a second denoises
x_t (B,2048,3); feature-transfer modules ball-query the condition branch's
multi-level features into the x_t branch at every encoder / decoder level.  During
sampling the condition branch (its SA/FP stacks and the global PointNet) depends
only on the partial cloud, so with use_retained_condition_feature=True it runs on
the first step only and its per-level outputs are retained for the remaining
T-1 steps (:364-369, 411-414, 453-455).

Attribute / parameter names follow the reference (including its `fc_lyaer`
spelling) so `load_state_dict(checkpoint['model_state_dict'])` works key for key.
"""
import copy

import torch
import torch.nn as nn

from ...pointnet2_ops.pointnet2_modules import FeatureMapModule, Swish
from .pnet import Pnet2Stage
from .pointnet2_ssg_sem import PointNet2SemSegSSG, calc_t_emb, swish


class PointNet2CloudCondition(PointNet2SemSegSSG):

    # ------------------------------------------------------------------ build
    def _feature_mapper(self, spec, radius, nsample, first_conv, first_conv_in, query_dim):
        hp = self.hparams
        return FeatureMapModule(
            spec, radius, nsample, use_xyz=hp["model.use_xyz"], include_abs_coordinate=self.include_abs_coordinate,
            include_center_coordinate=hp.get("include_center_coordinate", False), bn=self.bn,
            bn_first=hp["bn_first"], bias=hp["bias"], res_connect=hp["res_connect"], first_conv=first_conv,
            first_conv_in_channel=first_conv_in,
            neighbor_def=hp['feature_mapper_architecture']['neighbor_definition'],
            activation=self.network_activation, attention_setting=self.FeatureMapper_attention_setting,
            query_feature_dim=query_dim)

    def _build_model(self):
        hp = self.hparams
        self.reset_cond_features()
        if hp.get('concate_partial_with_noisy_input', False):
            raise NotImplementedError('single-network (concatenated input) variant is not on the built path')
        if hp.get('use_position_encoding', False):
            raise NotImplementedError('positional encoding is off in every shipped config and is not built')
        self.concate_partial_with_noisy_input = False
        self.use_position_encoding = False

        self.attention_setting = hp.get("attention_setting", None)
        self.FeatureMapper_attention_setting = copy.deepcopy(self.attention_setting)
        if self.FeatureMapper_attention_setting is not None:
            self.FeatureMapper_attention_setting['use_attention_module'] = \
                self.FeatureMapper_attention_setting['add_attention_to_FeatureMapper_module']
        self.global_attention_setting = hp.get('global_attention_setting', None)

        self.bn = hp.get("bn", True)  # "bn" means GroupNorm(32) throughout
        self.scale_factor = 1
        self.record_neighbor_stats = hp["record_neighbor_stats"]
        self.include_abs_coordinate = hp['include_abs_coordinate']
        self.pooling = hp.get('pooling', 'max')
        self.network_activation = hp.get('activation', 'relu')
        assert self.network_activation in ['relu', 'swish']
        self.include_local_feature = hp.get('include_local_feature', True)
        self.include_global_feature = hp.get('include_global_feature', False)
        self.attach_position_to_input_feature = hp['attach_position_to_input_feature']

        if hp["include_class_condition"]:
            self.class_emb = nn.Embedding(hp["num_class"], hp["class_condition_dim"])

        in_dim = hp['in_fea_dim']
        cond_in_dim = hp.get('partial_in_fea_dim', in_dim)
        if self.attach_position_to_input_feature:
            in_dim, cond_in_dim = in_dim + 3, cond_in_dim + 3
        self.partial_in_fea_dim = cond_in_dim

        self.global_feature_dim = None
        if self.include_global_feature:
            g_arch = hp['pnet_global_feature_architecture']
            self.global_feature_dim = g_arch[1][-1]
            self.global_pnet = Pnet2Stage(g_arch[0], g_arch[1], bn=self.bn,
                                          remove_last_activation=hp.get('global_feature_remove_last_activation',
                                                                        True))

        t_dim = hp['t_dim']
        self.fc_t1 = nn.Linear(t_dim, 4 * t_dim)
        self.fc_t2 = nn.Linear(4 * t_dim, 4 * t_dim)
        self.activation = swish

        arch = hp['architecture']
        feature_dim = arch['feature_dim']
        dec_dim = arch['decoder_feature_dim']
        assert dec_dim[-1] == feature_dim[-1]
        enc_map_dim = dec_map_dim = None

        if self.include_local_feature:
            c_arch = hp['condition_net_architecture']
            c_feat = c_arch['feature_dim']
            c_dec = c_arch['decoder_feature_dim']
            assert c_dec[-1] == c_feat[-1]
            m_arch = hp['feature_mapper_architecture']
            enc_map_dim, dec_map_dim = m_arch['encoder_feature_map_dim'], m_arch['decoder_feature_map_dim']

            self.SA_modules_condition = self.build_SA_model(
                c_arch['npoint'], c_arch['radius'], c_arch['nsample'], c_feat, c_arch['mlp_depth'], cond_in_dim,
                False, False, neighbor_def=c_arch['neighbor_definition'], activation=self.network_activation,
                bn=self.bn, attention_setting=self.attention_setting)

            self.encoder_feature_map = nn.ModuleList()
            for i, out_dim in enumerate(enc_map_dim):
                first_conv = hp["bn_first"] and i == 0
                src_dim = cond_in_dim if (i == 0 and not first_conv) else c_feat[i]
                query_dim = in_dim if i == 0 else feature_dim[i]
                self.encoder_feature_map.append(self._feature_mapper(
                    [src_dim] + [out_dim] * m_arch['encoder_mlp_depth'], m_arch['encoder_radius'][i],
                    m_arch['encoder_nsample'][i], first_conv, cond_in_dim, query_dim))

        self.SA_modules = self.build_SA_model(
            arch['npoint'], arch['radius'], arch['nsample'], feature_dim, arch['mlp_depth'],
            in_dim + enc_map_dim[0] if self.include_local_feature else in_dim,
            hp['include_t'], hp["include_class_condition"], include_global_feature=self.include_global_feature,
            global_feature_dim=self.global_feature_dim, additional_fea_dim=enc_map_dim,
            neighbor_def=arch['neighbor_definition'], activation=self.network_activation, bn=self.bn,
            attention_setting=self.attention_setting, global_attention_setting=self.global_attention_setting)

        if self.include_local_feature:
            self.FP_modules_condition = self.build_FP_model(
                c_dec, c_arch['decoder_mlp_depth'], c_feat, cond_in_dim, False, False,
                use_knn_FP=c_arch.get('use_knn_FP', False), K=c_arch.get('K', 3),
                include_grouper=c_arch.get('include_grouper', False), radius=c_arch['radius'],
                nsample=c_arch['nsample'], neighbor_def=c_arch['neighbor_definition'],
                activation=self.network_activation, bn=self.bn, attention_setting=self.attention_setting)

            self.decoder_feature_map = nn.ModuleList()
            for i, out_dim in enumerate(dec_map_dim):
                self.decoder_feature_map.append(self._feature_mapper(
                    [c_dec[i]] + [out_dim] * m_arch['decoder_mlp_depth'], m_arch['decoder_radius'][i],
                    m_arch['decoder_nsample'][i], False, 0, dec_dim[i]))

        self.FP_modules = self.build_FP_model(
            dec_dim, arch['decoder_mlp_depth'], feature_dim, in_dim, hp['include_t'],
            hp["include_class_condition"], include_global_feature=self.include_global_feature,
            global_feature_dim=self.global_feature_dim,
            additional_fea_dim=dec_map_dim[1:] if self.include_local_feature else None,
            use_knn_FP=arch.get('use_knn_FP', False), K=arch.get('K', 3),
            include_grouper=arch.get('include_grouper', False), radius=arch['radius'], nsample=arch['nsample'],
            neighbor_def=arch['neighbor_definition'], activation=self.network_activation, bn=self.bn,
            attention_setting=self.attention_setting, global_attention_setting=self.global_attention_setting)

        # refinement + upsampling: out_dim = 3 * (f [+1])   (:238-244)
        factor = hp.get('point_upsample_factor', 1)
        if factor > 1:
            if hp.get('include_displacement_center_to_final_output', False):
                factor -= 1
            hp['out_dim'] = int(hp['out_dim'] * (factor + 1))

        head_in = dec_dim[0] + 3 + (dec_map_dim[0] if self.include_local_feature else 0)
        act = nn.ReLU(True) if self.network_activation == 'relu' else Swish()
        if hp["bn_first"]:
            head = [act, nn.Conv1d(head_in, hp['out_dim'], kernel_size=1)]
        else:
            head = [nn.Conv1d(head_in, 128, kernel_size=1, bias=hp["bias"])]
            if self.bn:
                head.append(nn.GroupNorm(32, 128))
            head += [act, nn.Conv1d(128, hp['out_dim'], kernel_size=1)]
        self.fc_lyaer = nn.Sequential(*head)

    # ---------------------------------------------------------------- caching
    def reset_cond_features(self):
        self.l_uvw = None                  # condition xyz per level
        self.encoder_cond_features = None  # condition features per encoder level
        self.decoder_cond_features = None  # condition features per decoder level
        self.global_feature = None         # Pnet2Stage output

    # ---------------------------------------------------------------- forward
    def _embeddings(self, ts, label):
        hp = self.hparams
        t_emb = None
        if ts is not None and hp['include_t']:
            t_emb = self.activation(self.fc_t1(calc_t_emb(ts, hp['t_dim'])))
            t_emb = self.activation(self.fc_t2(t_emb))
        class_emb = None
        if label is not None and hp['include_class_condition']:
            class_emb = self.class_emb(label)
        return t_emb, class_emb

    def forward(self, pointcloud, condition, ts=None, label=None, use_retained_condition_feature=False):
        """pointcloud (B,N,3[+C]); condition (B,M,3[+C']); ts (B,) float steps or None;
        label (B,) long or None.  Returns (B,N,out_dim)."""
        hp = self.hparams
        if self.include_global_feature or self.include_local_feature:
            assert condition is not None
        retain = use_retained_condition_feature
        with torch.no_grad():
            if self.attach_position_to_input_feature:
                pointcloud = torch.cat([pointcloud, pointcloud[:, :, 0:3] / self.scale_factor], dim=2)
                if condition is not None:
                    condition = torch.cat([condition, condition[:, :, 0:3] / self.scale_factor], dim=2)
                raw_cond_dim = self.partial_in_fea_dim - 3
            else:
                raw_cond_dim = self.partial_in_fea_dim
            xyz, features = self._break_up_pc(pointcloud)
            xyz = xyz / self.scale_factor
            if condition is not None:
                uvw, cond_features = self._break_up_pc(condition)
                uvw = uvw / self.scale_factor

        t_emb, class_emb = self._embeddings(ts, label)

        if self.include_global_feature:
            if retain and self.global_feature is not None:
                global_feature = self.global_feature
            else:
                g_in = torch.cat([uvw, condition[:, :, 3:3 + raw_cond_dim]], dim=2) if raw_cond_dim > 0 else uvw
                global_feature = self.global_pnet(g_in.transpose(1, 2))
                if retain:
                    self.global_feature = global_feature.detach().clone()
            condition_emb = global_feature
            second_condition_emb = class_emb if hp['include_class_condition'] else None
        else:
            condition_emb = class_emb if hp['include_class_condition'] else None
            second_condition_emb = None

        stats, pool = self.record_neighbor_stats, self.pooling
        local = self.include_local_feature
        l_xyz, l_features = [xyz], [features]
        if condition is not None:
            l_uvw, l_cond = [uvw], [cond_features]

        # ------------------------------ encoder
        enc_cached = local and retain and self.encoder_cond_features is not None
        for i, sa in enumerate(self.SA_modules):
            if local:
                if enc_cached:
                    src_xyz, src_feat = self.l_uvw[i], self.encoder_cond_features[i]
                else:
                    nxt_uvw, nxt_feat = self.SA_modules_condition[i](
                        l_uvw[i], l_cond[i], t_emb=None, condition_emb=None, subset=True,
                        record_neighbor_stats=stats, pooling=pool)
                    l_uvw.append(nxt_uvw)
                    l_cond.append(nxt_feat)
                    src_xyz, src_feat = l_uvw[i], l_cond[i]
                mapped = self.encoder_feature_map[i](src_xyz, src_feat, l_xyz[i], subset=False,
                                                     record_neighbor_stats=stats, pooling=pool,
                                                     features_at_new_xyz=l_features[i])
                sa_in = torch.cat([mapped, l_features[i]], dim=1)
            else:
                sa_in = l_features[i]
            nxt_xyz, nxt_features = sa(l_xyz[i], sa_in, t_emb=t_emb, condition_emb=condition_emb,
                                       second_condition_emb=second_condition_emb, subset=True,
                                       record_neighbor_stats=stats, pooling=pool)
            l_xyz.append(nxt_xyz)
            l_features.append(nxt_features)
        if local and retain and self.l_uvw is None:
            self.l_uvw = l_uvw
            self.encoder_cond_features = copy.deepcopy(l_cond)

        # ------------------------------ decoder
        dec_cached = local and retain and self.decoder_cond_features is not None
        for i in range(-1, -(len(self.FP_modules) + 1), -1):
            if local:
                if dec_cached:
                    src_xyz, src_feat = self.l_uvw[i], self.decoder_cond_features[i]
                else:
                    l_cond[i - 1] = self.FP_modules_condition[i](
                        l_uvw[i - 1], l_uvw[i], l_cond[i - 1], l_cond[i], t_emb=None, condition_emb=None,
                        record_neighbor_stats=stats, pooling=pool)
                    src_xyz, src_feat = l_uvw[i], l_cond[i]
                mapped = self.decoder_feature_map[i](src_xyz, src_feat, l_xyz[i], subset=False,
                                                     record_neighbor_stats=stats, pooling=pool,
                                                     features_at_new_xyz=l_features[i])
                fp_in = torch.cat([mapped, l_features[i]], dim=1)
            else:
                fp_in = l_features[i]
            l_features[i - 1] = self.FP_modules[i](
                l_xyz[i - 1], l_xyz[i], l_features[i - 1], fp_in, t_emb=t_emb, condition_emb=condition_emb,
                second_condition_emb=second_condition_emb, record_neighbor_stats=stats, pooling=pool)

        if local:
            if retain and self.decoder_cond_features is None:
                self.decoder_cond_features = copy.deepcopy(l_cond)
            if retain:
                src_xyz, src_feat = self.l_uvw[0], self.decoder_cond_features[0]
            else:
                src_xyz, src_feat = l_uvw[0], l_cond[0]
            mapped = self.decoder_feature_map[0](src_xyz, src_feat, l_xyz[0], subset=False,
                                                 record_neighbor_stats=stats, pooling=pool,
                                                 features_at_new_xyz=l_features[0])
            out_feature = torch.cat([mapped, l_features[0]], dim=1)
        else:
            out_feature = l_features[0]

        out_feature = torch.cat([out_feature, xyz.transpose(1, 2)], dim=1)
        return torch.transpose(self.fc_lyaer(out_feature), 1, 2)
