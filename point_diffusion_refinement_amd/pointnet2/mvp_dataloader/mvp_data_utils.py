"""Point-cloud augmentation of the MVP reader (reference pointnet2/mvp_dataloader/mvp_data_utils.py:8-66).

Generation can run on augmented inputs (`augment_data_during_generation`, completion_eval.py:140-143): the
dataset then hands out the inverse transform (`M_inv`, `translation`) with every item and the harness maps the
completed cloud back (completion_eval.py:203-211; here generation.evaluate_batch).  This module builds the same
transform from the same draws in the same order -- `random.uniform` (scale), `random.uniform` (rotation about y),
two `random.random` (mirror x / mirror z), `np.random.normal` (translation), `np.random.randn` (jitter) -- so a
seeded run is reproducible against the reference; the 3 x 3 factors are written out instead of going through
transforms3d (uniform zoom s I, reflection I - 2 n n^T, Rodrigues rotation about y).
"""
import math
import random

import numpy as np


def _zoom(s, direction=None):
    if direction is None:
        return np.eye(3) * s
    n = np.asarray(direction, dtype=np.float64)
    n = n / np.linalg.norm(n)
    return np.eye(3) - (1.0 - s) * np.outer(n, n)


def _rot_y(angle):
    c, s = math.cos(angle), math.sin(angle)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])


def augment_cloud(Ps, args, return_augmentation_params=False):
    """Ps: list of (N, >=3) arrays sharing ONE random transform; the xyz columns are transformed in place like the
    reference does.  args: pc_augm_scale, pc_augm_rot, pc_rot_scale (degrees), pc_augm_mirror_prob, pc_augm_jitter,
    optional translation_magnitude."""
    M = _zoom(1)
    if args['pc_augm_scale'] > 1:
        s = random.uniform(1 / args['pc_augm_scale'], args['pc_augm_scale'])
        M = np.dot(_zoom(s), M)
    if args['pc_augm_rot']:
        scale = args['pc_rot_scale']
        if scale > 0:
            angle = random.uniform(-math.pi, math.pi) * scale / 180.0
            M = np.dot(_rot_y(angle), M)                      # y is the upright axis of the MVP shapes
    if args['pc_augm_mirror_prob'] > 0:                       # mirror x and / or z, never y
        if random.random() < args['pc_augm_mirror_prob'] / 2:
            M = np.dot(_zoom(-1, [1, 0, 0]), M)
        if random.random() < args['pc_augm_mirror_prob'] / 2:
            M = np.dot(_zoom(-1, [0, 0, 1]), M)
    translation_sigma = max(args['pc_augm_scale'], 1) * args.get('translation_magnitude', 0)
    noise = None
    if translation_sigma > 0:
        noise = np.random.normal(scale=translation_sigma, size=(1, 3)).astype(Ps[0].dtype)
    result = []
    for P in Ps:
        P[:, :3] = np.dot(P[:, :3], M.T)
        if noise is not None:
            P[:, :3] = P[:, :3] + noise
        if args['pc_augm_jitter']:
            sigma, clip = 0.01, 0.05
            P = P + np.clip(sigma * np.random.randn(*P.shape), -1 * clip, clip).astype(np.float32)
        result.append(P)
    if return_augmentation_params:
        params = {'M_inv': np.linalg.inv(M.T).astype(Ps[0].dtype),
                  'translation': noise if noise is not None else np.zeros((1, 3)).astype(Ps[0].dtype)}
        return result, params
    return result
