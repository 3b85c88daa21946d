"""Seeded inputs of the data-path goldens, shared by the fixture generator (tests/golden/make_golden.py dataset(),
which runs the REFERENCE reader over them) and tests/test_data_path.py (which runs the product's reader)."""
import os

import numpy as np
import torch

G_NORMAL, G_NOVEL, N_PARTIAL, N_GT, N_MIRROR = 5, 2, 16, 32, 24
AUG = {"pc_augm_scale": 1.2, "pc_augm_rot": True, "pc_rot_scale": 90.0, "pc_augm_mirror_prob": 0.5,
       "pc_augm_jitter": False, "translation_magnitude": 0.1}

# (name, ShapeNetH5 keyword arguments, seed of `random` / `np.random` before construction)
CASES = [
    ("test_w2_r0", dict(train=False, npoints=N_GT, rank=0, world_size=2, append_samples_to_last_rank=False), 1),
    ("test_w2_r1", dict(train=False, npoints=N_GT, rank=1, world_size=2, append_samples_to_last_rank=False), 2),
    ("test_w3_r2_scale", dict(train=False, npoints=N_GT, rank=2, world_size=3, scale=0.5,
                              append_samples_to_last_rank=False), 3),
    ("test_mirrored", dict(train=False, npoints=N_GT, use_mirrored_partial_input=True, number_partial_points=N_MIRROR,
                           scale=1.5), 4),
    ("train_w2_r1_append", dict(train=True, npoints=N_GT, rank=1, world_size=2, append_samples_to_last_rank=True), 5),
    ("test_subsample", dict(train=False, npoints=N_GT, random_subsample=True, num_samples=40), 6),
    ("test_novel_only", dict(train=False, npoints=N_GT, novel_input_only=True), 7),
    ("test_augmented", dict(train=False, npoints=N_GT, augmentation=AUG, return_augmentation_params=True), 8),
]


def item_indices(n):
    return sorted({0, min(27, n - 1), n - 1})


def source_arrays():
    rng = np.random.default_rng(2024)
    out = {}
    for split, ofs in (("test", 0), ("train", 100)):
        gt = (rng.random((G_NORMAL, N_GT, 3)) - 0.5).astype(np.float32)
        ngt = (rng.random((G_NOVEL, N_GT, 3)) - 0.5).astype(np.float32)
        def views(g):
            return np.stack([c[rng.permutation(N_GT)[:N_PARTIAL]] for c in g for _ in range(26)]).astype(np.float32)
        out[split + "_incomplete_pcds"], out[split + "_novel_incomplete_pcds"] = views(gt), views(ngt)
        out[split + "_labels"] = np.repeat((np.arange(G_NORMAL) + ofs) % 8, 26).astype(np.int64)
        out[split + "_novel_labels"] = np.repeat(8 + np.arange(G_NOVEL), 26).astype(np.int64)
        out[split + "_complete_pcds"], out[split + "_novel_complete_pcds"] = gt, ngt
        mir = rng.random(((G_NORMAL + G_NOVEL) * 26, N_MIRROR, 4)).astype(np.float32) - 0.5
        mir[:, :, 3] = np.where(np.arange(N_MIRROR) < N_MIRROR // 2, 1.0, -1.0)
        out[split + "_mirrored"] = mir
    return out


def write_directory(root, src, write):
    """The reference's file layout (mvp_dataset.py:45-57); write(path, {name: array})."""
    os.makedirs(os.path.join(root, "mirror_and_concated_partial"), exist_ok=True)
    for split in ("test", "train"):
        write(os.path.join(root, "mvp_%s_input.h5" % split),
              {k: src["%s_%s" % (split, k)] for k in ("incomplete_pcds", "labels", "novel_incomplete_pcds", "novel_labels")})
        write(os.path.join(root, "mvp_%s_gt_%dpts.h5" % (split, N_GT)),
              {k: src["%s_%s" % (split, k)] for k in ("complete_pcds", "novel_complete_pcds")})
        write(os.path.join(root, "mirror_and_concated_partial", "mvp_%s_input_mirror_and_concat_%dpts.h5" % (split, N_MIRROR)),
              {"data": src[split + "_mirrored"]})


def deaugment_inputs():
    g = torch.Generator().manual_seed(77)
    gen, gt = torch.randn(3, 20, 3, generator=g), torch.randn(3, 20, 3, generator=g)
    M = torch.linalg.qr(torch.randn(3, 3, 3, generator=g)).Q * torch.tensor([1.1, 0.9, 1.3])[:, None, None]
    return gen, gt, torch.linalg.inv(M.transpose(1, 2)).contiguous(), 0.1 * torch.randn(3, 1, 3, generator=g)
