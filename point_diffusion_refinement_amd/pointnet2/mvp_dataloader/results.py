"""Per-rank result files of a generation job and their gatherer -- same files, names and contents as the
reference writes (completion_eval.py:267-296: `mvp_generated_data_<n>pts.h5` with dataset 'data';
generate_samples.py: one evaluation pickle per rank) and reads back (generate_samples_distributed.py:26-97
`gather_generated_results`): rank directories `rank_<r>` under one father directory, gathered by plain
concatenation IN RANK ORDER, means over the concatenated arrays.

The RCCL all-gather of generation.gather_records produces the same (sum n_r, 5) table in memory; this module is
the on-disk form for jobs whose clouds must be kept (they feed the refinement network's training set,
mvp_dataset.py:107-127 `include_generated_samples`).
"""
import os
import pickle

import numpy as np

from .shard_io import load_array, save_arrays


def rank_dir(father_directory, rank):
    d = os.path.join(father_directory, "rank_%d" % rank)
    os.makedirs(d, exist_ok=True)
    return d


def save_rank_results(father_directory, rank, generated, records, iteration=0, num_points=None,
                      eval_file="mvp_eval_result.pkl", dataset="mvp"):
    """generated: (n, N, 3) clouds of this rank (already / 2 / scale, as evaluate_batch returns them) or None;
    records: (n, 5) [cd_t, cd_p, f1, emd, label] (generation.evaluate_batch)."""
    d = rank_dir(father_directory, rank)
    rec = np.asarray(records, dtype=np.float32).reshape(-1, 5)
    paths = []
    if generated is not None:
        g = np.asarray(generated, dtype=np.float32)
        n = g.shape[1] if num_points is None else num_points
        paths.append(save_arrays(os.path.join(d, "%s_generated_data_%dpts.h5" % (dataset, n)), {"data": g}))
    with open(os.path.join(d, eval_file), "wb") as f:
        pickle.dump({"meta": rec[:, 4].astype(np.int64), "cd_distance": rec[:, 0], "cd_p": rec[:, 1],
                     "emd_distance": rec[:, 3], "f1": rec[:, 2],
                     "avg_cd": float(rec[:, 0].mean()) if len(rec) else 0.0,
                     "avg_emd": float(rec[:, 3].mean()) if len(rec) else 0.0, "iter": iteration}, f)
    paths.append(os.path.join(d, eval_file))
    return paths


def gather_generated_results(father_directory, num_ranks, remove_original_files=False):
    """Reference generate_samples_distributed.py:26-97.  Returns the gathered evaluation dict."""
    data, meta, cd, emd, f1 = {}, [], [], [], []
    iteration, eval_save_file = None, None
    log = []
    for rank in range(num_ranks):
        directory = os.path.join(father_directory, "rank_%d" % rank)
        for fl in sorted(os.listdir(directory)):
            file_name = os.path.join(directory, fl)
            if fl.endswith(".h5") or fl.endswith(".npz"):
                arr = load_array(file_name, "data")
                data.setdefault(fl, []).append(arr)
                log.append("data from %s is of shape %s" % (file_name, arr.shape))
            elif fl.endswith(".pkl"):
                eval_save_file = fl
                with open(file_name, "rb") as h:
                    r = pickle.load(h)
                meta.append(r["meta"])
                cd.append(r["cd_distance"])
                emd.append(r["emd_distance"])
                f1.append(r["f1"])
                iteration = r["iter"]
            else:
                continue
            if remove_original_files:
                os.remove(file_name)
                log.append("%s is removed" % file_name)
    for key, parts in data.items():
        arr = np.concatenate(parts, axis=0)
        out = save_arrays(os.path.join(father_directory, key), {"data": arr})
        log.append("The gathered data from all %s files of different ranks is of shape %s, saved to %s"
                   % (key, arr.shape, out))
    meta, cd, emd, f1 = (np.concatenate(x, axis=0) for x in (meta, cd, emd, f1))
    gathered = {"meta": meta, "cd_distance": cd, "emd_distance": emd, "f1": f1, "avg_cd": cd.mean(),
                "avg_emd": emd.mean(), "iter": iteration}
    with open(os.path.join(father_directory, eval_save_file), "wb") as h:
        pickle.dump(gathered, h)
    log.append("CD loss: {} EMD loss: {} F1 Score: {}".format(cd.mean(), emd.mean(), f1.mean()))
    with open(os.path.join(father_directory, "gathered_generation.log"), "w") as h:
        h.write("\n".join(log) + "\n")
    return gathered
