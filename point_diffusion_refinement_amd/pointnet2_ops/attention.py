"""Per-neighbourhood attention pooling (reference pointnet2_ops/attention.py:35-96).

`AttentionModule` replaces max-pooling over the K neighbours of every SA /
feature-transfer / kNN-FP block: scores = MLP([conv(query) | conv(key)]) masked by
the ball-query count, softmax over K, weighted sum of conv(value).  Parameter
names (feat_conv, grouped_feat_conv, weight_conv.{1,2,4,5}, feat_out_conv.{0,1})
match the reference so its checkpoints load key for key.  `GlobalAttentionModule`
(N x N, attention.py:98-154) is not enabled by any shipped config and is not built.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class MyGroupNorm(nn.Module):
    """GroupNorm over the first C - C % G channels; trailing channels (appended xyz
    coordinates) pass through un-normalised (attention.py:6-23)."""

    def __init__(self, num_groups, num_channels):
        super().__init__()
        self.num_groups = num_groups
        self.num_channels = num_channels - num_channels % num_groups
        self.group_norm = nn.GroupNorm(self.num_groups, self.num_channels)

    def forward(self, x):
        c = self.num_channels
        if x.shape[1] == c:
            return self.group_norm(x)
        return torch.cat([self.group_norm(x[:, :c]), x[:, c:]], dim=1)


def count_to_mask(count, K):
    ar = torch.arange(K, device=count.device, dtype=count.dtype)
    return ar.view(1, 1, K) < count.unsqueeze(-1)


def _act_norm_conv(cin, cout, norm):
    layers = [nn.ReLU(inplace=True)]
    if norm:
        layers.append(MyGroupNorm(min(32, cin), cin))
    layers.append(nn.Conv2d(cin, cout, kernel_size=1))
    return layers


class AttentionModule(nn.Module):
    def __init__(self, C_in1, C_in2, C1, C2, C_out, attention_bn=True, transform_grouped_feat_out=True,
                 last_activation=True):
        super().__init__()
        C1, C2 = max(C1, 32), max(C2, 32)
        self.feat_conv = nn.Conv2d(C_in1, C1, kernel_size=1)
        self.grouped_feat_conv = nn.Conv2d(C_in2, C2, kernel_size=1)
        inter = min(C1 + C2, C_out)
        self.weight_conv = nn.Sequential(*(_act_norm_conv(C1 + C2, inter, attention_bn) +
                                           _act_norm_conv(inter, C_out, attention_bn)))
        self.transform_grouped_feat_out = transform_grouped_feat_out
        if transform_grouped_feat_out:
            tail = [nn.Conv2d(C_out, C_out, kernel_size=1)]
            if last_activation:
                if attention_bn:
                    tail.append(MyGroupNorm(min(32, C_out), C_out))
                tail.append(nn.ReLU(inplace=True))
            self.feat_out_conv = nn.Sequential(*tail)

    def forward(self, feat, grouped_feat, grouped_feat_out, count):
        """feat (B,C_in1,N) query; grouped_feat (B,C_in2,N,K) key; grouped_feat_out (B,C_out,N,K)
        value; count (B,N) valid neighbours or 'all'.  Returns (B,C_out,N)."""
        K = grouped_feat.shape[-1]
        q = self.feat_conv(feat.unsqueeze(-1)).expand(-1, -1, -1, K)
        k = self.grouped_feat_conv(grouped_feat)
        scores = self.weight_conv(torch.cat([q, k], dim=1))
        if not (isinstance(count, str) and count == 'all'):
            mask = count_to_mask(torch.clamp(count, min=1), K).unsqueeze(1).float()
            scores = scores * mask + (-1e9) * (1 - mask)
        weight = F.softmax(scores, dim=-1)
        if self.transform_grouped_feat_out:
            grouped_feat_out = self.feat_out_conv(grouped_feat_out)
        return (grouped_feat_out * weight).sum(dim=-1)
