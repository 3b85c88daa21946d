"""MVP completion dataset -- the reader of reference pointnet2/mvp_dataloader/mvp_dataset.py:16-328 for the
generation / evaluation path.

Same file layout under `data_dir` (datasets named as in the reference's h5 files):
    mvp_{train,test}_input.h5            incomplete_pcds (P,2048,3), labels (P,), novel_incomplete_pcds, novel_labels
    mvp_{train,test}_gt_<npoints>pts.h5  complete_pcds (G,npoints,3), novel_complete_pcds          (P = 26 G)
    mirror_and_concated_partial/mvp_{train,test}_input_mirror_and_concat_<n>pts.h5   data (P,n,4)
Each file may also be an '.npz' with the same array names (shard_io.resolve).

Semantics kept from the reference:
  * normal + novel categories concatenated (:141-157; `novel_input_only` keeps the novel ones);
  * 26 partial views per complete shape, ground truth of partial i is shape i // 26 (:289);
  * multi-rank split by COMPLETE shapes: per = ceil(G / W), rank r owns shapes [r per, (r+1) per) and partials
    [26 r per, 26 (r+1) per) (:160-198); with `append_samples_to_last_rank` (training only) the short last rank is
    topped up with randomly drawn shapes (:171-196) -- generation passes False (generate_samples.py:191-192);
  * coordinates scaled by 2 * scale (the files hold [-0.5, 0.5]; mirrored files: xyz only, 4th channel = tag) (:250-262);
  * items are dicts {'partial', 'complete', 'label'} (+ 'generated', 'XT' when requested);
  * `augmentation` (a dict, mvp_data_utils.augment_cloud) transforms every cloud of an item with ONE random
    transform (:292-313); with `return_augmentation_params` the item carries 'M_inv' and 'translation' so that
    the harness can map generated clouds back (completion_eval.py:140-143, 203-211).
"""
import os
import random
import warnings

import numpy as np
import torch
import torch.utils.data as data

from .mvp_data_utils import augment_cloud
from .shard_io import load_array

VIEWS = 26


class ShapeNetH5(data.Dataset):
    def __init__(self, data_dir, train=True, npoints=2048, novel_input=True, novel_input_only=False, scale=1, rank=0,
                 world_size=1, random_subsample=False, num_samples=1000, augmentation=False,
                 return_augmentation_params=False, include_generated_samples=False, generated_sample_path=None,
                 randomly_select_generated_samples=False, use_mirrored_partial_input=False,
                 number_partial_points=2048, load_pre_computed_XT=False, T_step=100, XT_folder=None,
                 append_samples_to_last_rank=True):
        self.augmentation, self.return_augmentation_params = augmentation, return_augmentation_params
        if use_mirrored_partial_input or load_pre_computed_XT:
            assert novel_input and not novel_input_only
        split = "train" if train else "test"
        self.use_mirrored_partial_input = use_mirrored_partial_input
        self.input_path = os.path.join(data_dir, "mvp_%s_input.h5" % split)
        self.gt_path = os.path.join(data_dir, "mvp_%s_gt_%dpts.h5" % (split, npoints))
        self.npoints, self.train = npoints, train

        input_data = load_array(self.input_path, "incomplete_pcds")
        labels = load_array(self.input_path, "labels")
        novel_input_data = load_array(self.input_path, "novel_incomplete_pcds")
        novel_labels = load_array(self.input_path, "novel_labels")
        gt_data = load_array(self.gt_path, "complete_pcds")
        novel_gt_data = load_array(self.gt_path, "novel_complete_pcds")

        self.load_pre_computed_XT = load_pre_computed_XT
        generated_XT = None
        if load_pre_computed_XT:
            xt_file = os.path.join(XT_folder, split, "mvp_generated_data_2048pts_T%d.h5" % T_step)
            generated_XT = load_array(xt_file, "data")
        self.include_generated_samples = include_generated_samples
        generated = None
        if include_generated_samples:
            gdir = os.path.join(data_dir, generated_sample_path)
            if randomly_select_generated_samples:
                trials = [os.path.join(gdir, f) for f in sorted(os.listdir(gdir)) if f.startswith("trial")]
                gdir = random.choice([gdir] + trials)
            generated = load_array(os.path.join(gdir, split, "mvp_generated_data_2048pts.h5"), "data")

        if novel_input_only:
            input_data, gt_data, labels = novel_input_data, novel_gt_data, novel_labels
        elif novel_input:
            if use_mirrored_partial_input:
                self.mirrored_input_path = os.path.join(
                    data_dir, "mirror_and_concated_partial",
                    "mvp_%s_input_mirror_and_concat_%dpts.h5" % (split, number_partial_points))
                input_data = load_array(self.mirrored_input_path, "data")
            else:
                input_data = np.concatenate((input_data, novel_input_data), axis=0)
            gt_data = np.concatenate((gt_data, novel_gt_data), axis=0)
            labels = np.concatenate((labels, novel_labels), axis=0)

        if world_size > 1:
            G = gt_data.shape[0]
            per = int(np.ceil(G / world_size))
            start, end = rank * per, (rank + 1) * per
            supp = None
            if rank == world_size - 1 and append_samples_to_last_rank and end * VIEWS - input_data.shape[0] > 0:
                assert train, "samples are appended to the last rank only when training"
                missing = end - G
                supp_gt = np.array(random.sample(list(range(G)), missing))
                supp_partial = (supp_gt[:, None] * VIEWS + np.arange(VIEWS)[None]).reshape(-1)
                supp = (input_data[supp_partial], labels[supp_partial], gt_data[supp_gt],
                        generated[supp_partial] if generated is not None else None,
                        generated_XT[supp_partial] if generated_XT is not None else None)
            input_data = input_data[start * VIEWS:end * VIEWS]
            gt_data = gt_data[start:end]
            labels = labels[start * VIEWS:end * VIEWS]
            if generated is not None:
                generated = generated[start * VIEWS:end * VIEWS]
            if generated_XT is not None:
                generated_XT = generated_XT[start * VIEWS:end * VIEWS]
            if supp is not None:
                input_data = np.concatenate([input_data, supp[0]], 0)
                labels = np.concatenate([labels, supp[1]], 0)
                gt_data = np.concatenate([gt_data, supp[2]], 0)
                if generated is not None:
                    generated = np.concatenate([generated, supp[3]], 0)
                if generated_XT is not None:
                    generated_XT = np.concatenate([generated_XT, supp[4]], 0)

        self.random_subsample = random_subsample
        if random_subsample:
            if num_samples < input_data.shape[0]:
                p2c = np.repeat(np.arange(gt_data.shape[0]), VIEWS)
                idx = np.array(random.sample(list(range(input_data.shape[0])), num_samples))
                input_data, labels = input_data[idx], labels[idx]
                self.partial_to_complete_index = p2c[idx]
                if generated is not None:
                    generated = generated[idx]
                if generated_XT is not None:
                    generated_XT = generated_XT[idx]
            else:
                self.random_subsample = False
                warnings.warn("num_samples (%d) is not less than the number of shapes (%d): no subsampling"
                              % (num_samples, input_data.shape[0]))

        self.scale = scale
        input_data = np.array(input_data, dtype=np.float32)
        if use_mirrored_partial_input:
            input_data[:, :, 0:3] = input_data[:, :, 0:3] * 2 * scale          # 4th channel: the mirror tag
        else:
            input_data = input_data * 2 * scale
        self.input_data = input_data
        self.gt_data = np.array(gt_data, dtype=np.float32) * 2 * scale
        self.generated_sample = None if generated is None else np.array(generated, dtype=np.float32) * 2 * scale
        self.generated_XT = None if generated_XT is None else np.array(generated_XT, dtype=np.float32) * 2 * scale
        self.labels = labels.astype(int)
        self.len = self.input_data.shape[0]

    def __len__(self):
        return self.len

    def __getitem__(self, index):
        gt_idx = self.partial_to_complete_index[index] if self.random_subsample else index // VIEWS
        result = {"partial": self.input_data[index].copy(), "complete": self.gt_data[gt_idx].copy()}
        if self.generated_sample is not None:
            result["generated"] = self.generated_sample[index].copy()
        if self.generated_XT is not None:
            result["XT"] = self.generated_XT[index].copy()
        params = {}
        if isinstance(self.augmentation, dict):
            # one transform for every cloud of the item, in dict order (partial, complete, generated, XT)
            clouds = list(result.values())
            if self.return_augmentation_params:
                clouds, params = augment_cloud(clouds, self.augmentation, return_augmentation_params=True)
            else:
                clouds = augment_cloud(clouds, self.augmentation, return_augmentation_params=False)
            for key, cloud in zip(list(result.keys()), clouds):
                result[key] = cloud
            if self.generated_sample is not None:
                sigma = self.augmentation.get("noise_magnitude_for_generated_samples", 0)
                if sigma > 0:       # noise on the coarse cloud the refinement network is trained on (:305-313)
                    noise = np.random.normal(scale=sigma, size=result["generated"].shape)
                    result["generated"] = result["generated"] + noise.astype(result["generated"].dtype)
        result.update(params)
        result = {k: torch.from_numpy(v) for k, v in result.items()}
        result["label"] = self.labels[index]
        return result

    def batch(self, lo, hi, device=None):
        """(condition, label, gt) of THIS RANK's partial indices [lo, hi): feeds generation.evaluate_batch
        (gt expanded per view, gt_idx = index // 26)."""
        idx = np.arange(lo, hi)
        gt_idx = self.partial_to_complete_index[idx] if self.random_subsample else idx // VIEWS
        cond = torch.from_numpy(self.input_data[lo:hi])
        label = torch.from_numpy(self.labels[lo:hi].astype(np.int64))
        gt = torch.from_numpy(self.gt_data[gt_idx])
        if device is not None:
            cond, label, gt = cond.to(device), label.to(device), gt.to(device)
        return cond, label, gt
