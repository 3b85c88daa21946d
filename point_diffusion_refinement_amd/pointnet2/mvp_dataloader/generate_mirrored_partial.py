"""Producer of the mirrored condition clouds (reference mvp_dataloader/generate_mirrored_partial.py:16-76):
every partial cloud of a split -> mirror about z, tag, concatenate, FPS down to 2048 and 3072 points
(data_utils/mirror_partial.mirror_and_concat, on libpdr_hip.so), written as
    <data_dir>/mirror_and_concated_partial/mvp_{train,test}_input_mirror_and_concat_{4096,2048,3072}pts.h5   ('data')
in the files' own [-0.5, 0.5] range (the reference reads the dataset with scale = 0.5, i.e. x 2 x 0.5 = 1).

    python -m point_diffusion_refinement_amd.pointnet2.mvp_dataloader.generate_mirrored_partial DATA_DIR [--train]
"""
import argparse
import os

import numpy as np
import torch

from ..data_utils.mirror_partial import mirror_and_concat
from .mvp_dataset import ShapeNetH5
from .shard_io import save_arrays


def build_mirrored_partials(data_dir, train=False, batch_size=128, device=None, num_points=(2048, 3072), npoints=2048):
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    ds = ShapeNetH5(data_dir, train=train, npoints=npoints, novel_input=True, novel_input_only=False, scale=0.5)
    chunks = None
    with torch.no_grad():
        for lo in range(0, len(ds), batch_size):
            partial = torch.from_numpy(ds.input_data[lo:lo + batch_size]).to(device)
            outs = [t.cpu().numpy() for t in mirror_and_concat(partial, axis=2, num_points=list(num_points))]
            chunks = [[o] for o in outs] if chunks is None else [c + [o] for c, o in zip(chunks, outs)]
    out_dir = os.path.join(data_dir, "mirror_and_concated_partial")
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for c in chunks:
        arr = np.concatenate(c, 0)
        name = "mvp_%s_input_mirror_and_concat_%dpts.h5" % ("train" if train else "test", arr.shape[1])
        paths.append(save_arrays(os.path.join(out_dir, name), {"data": arr}))
    return paths


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("data_dir")
    ap.add_argument("--train", action="store_true")
    a = ap.parse_args()
    for p in build_mirrored_partials(a.data_dir, train=a.train):
        print("wrote", p)
