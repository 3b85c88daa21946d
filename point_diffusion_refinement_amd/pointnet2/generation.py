"""Batch-sharded test-set generation + evaluation across the GPUs of one node.

This is the arithmetic contract of the reference's generation harness -- not its CLI / file
plumbing -- for one process per GPU:

  * shard split      mvp_dataloader/mvp_dataset.py:152-198 : GT shapes are split evenly,
                     per = ceil(G / W); rank r owns shapes [r*per, (r+1)*per) and their 26 partial
                     views [r*per*26, (r+1)*per*26)  (the last rank may be short: generation uses
                     append_samples_to_last_rank=False, generate_samples.py:191-192)
  * per batch        completion_eval.py:145-265 : reset condition cache, T-step reverse sampling,
                     x/2/scale and gt/2/scale, Chamfer (cd_t, cd_p, F1 at 1e-4) and EMD per sample
  * gather           generate_samples_distributed.py:26-97 : per-rank results concatenated IN RANK
                     ORDER, then plain means.  The reference does this through .pkl/.h5 files; here
                     it is ONE all_gather of (n_local, 5) float32 records
                     [cd_t, cd_p, f1, emd, label] over RCCL (backend "nccl"; "gloo" in CPU tests).
                     No collective runs inside the sampling loop; generated clouds stay rank-local.
"""
import math

import torch
import torch.distributed as dist

VIEWS_PER_SHAPE = 26   # mvp_dataset.py:150,289: 26 partial views per complete shape


def rank_shard(num_shapes, rank, world_size, views=VIEWS_PER_SHAPE):
    """(first_partial, last_partial_exclusive, first_shape, last_shape_exclusive) owned by `rank`."""
    per = int(math.ceil(num_shapes / world_size)) if world_size > 1 else num_shapes
    start, end = rank * per, min((rank + 1) * per, num_shapes)
    start = min(start, num_shapes)
    return start * views, end * views, start, end


def batches(first, last, batch_size):
    """Consecutive [lo, hi) index ranges of a shard, the last one possibly short (no drop_last)."""
    return [(lo, min(lo + batch_size, last)) for lo in range(first, last, batch_size)]


def evaluate_batch(generate, condition, label, gt, scale=1.0, f1_threshold=1e-4, compute_emd=True, M_inv=None,
                   translation=None):
    """One batch of the harness: `generate(condition, label)` -> (B,N,3) completed clouds.
    M_inv (B,3,3) / translation (B,1,3): the inverse of the augmentation the dataset applied to condition and gt
    (`augment_data_during_generation`, completion_eval.py:140-143): generated clouds and gt are mapped back with
    `matmul(x - translation, M_inv)` before anything is measured (:203-205).
    Returns (generated/2/scale, records (B,5) = [cd_t, cd_p, f1, emd, label])."""
    from .chamfer_loss_new import calc_cd
    from .emd import earth_mover_distance
    generated = generate(condition, label)
    if M_inv is not None:
        shift = translation if translation is not None else torch.zeros_like(gt[:, :1])
        generated = torch.matmul(generated - shift, M_inv)
        gt = torch.matmul(gt - shift, M_inv)
    generated = generated / 2 / scale
    gt = gt / 2 / scale
    cd_p, cd_t, f1 = calc_cd(generated, gt, calc_f1=True, f1_threshold=f1_threshold)
    emd = earth_mover_distance(generated, gt) if compute_emd else torch.zeros_like(cd_t)
    rec = torch.stack([cd_t, cd_p, f1, emd, label.to(cd_t.dtype)], dim=1)
    return generated, rec


def refine_completion(refine_net, generated, condition, label, output_scale_factor, point_upsample_factor=1,
                      include_displacement_center_to_final_output=False):
    """Second stage of the paper (completion_eval.py:159-168, task 'refine_completion'): ONE forward of the
    refinement network (include_t = False) predicts a displacement of the coarse cloud; with
    point_upsample_factor f > 1 every coarse point emits f points (models/point_upsample_module.py:4-28).
    (B,N,3) coarse -> (B, N f, 3)."""
    from .models.point_upsample_module import point_upsample
    refine_net.reset_cond_features()
    # the fused network evaluates a ball that holds one point once (fused_network.DEDUP) -- the rule on the noisy x_t of
    # a reverse process, the exception HERE: the refinement input is a finished surface, every ball is full, the
    # per-query chain would be pure overhead (bench.py `trajectory.refinement_forward` times both).  One eager forward
    # per batch: no probe, the switch is simply off.
    saved = getattr(refine_net, "dedup", None)
    if saved is not None:
        refine_net.dedup = False
    try:
        displacement = refine_net(generated, condition, ts=None, label=label)
    finally:
        if saved is not None:
            refine_net.dedup = saved
    if point_upsample_factor > 1:
        refined, _ = point_upsample(generated, displacement, point_upsample_factor,
                                    include_displacement_center_to_final_output, output_scale_factor)
        return refined
    return generated + displacement * output_scale_factor


class GraphedRefiner:
    """`refine_completion` as ONE hipGraph replay per batch (round 6): the refinement forward is ~650 launches whose
    submission, not whose execution, sets its 15-24 ms when the host issues them one by one.  Captured at the first
    batch of a shape (after an eager warm-up call whose result is returned for that batch); later batches copy their
    inputs into the captured buffers and replay.  Same kernels, same results as the eager call."""

    def __init__(self, refine_net, output_scale_factor, point_upsample_factor=1,
                 include_displacement_center_to_final_output=False):
        self.net = refine_net
        self.args = (output_scale_factor, point_upsample_factor, include_displacement_center_to_final_output)
        self._key = None

    @torch.no_grad()
    def __call__(self, generated, condition, label):
        key = (tuple(generated.shape), tuple(condition.shape), None if label is None else tuple(label.shape),
               generated.device)
        if not generated.is_cuda or not hasattr(self.net, "sync_condition"):   # (the fused network's launches only)
            return refine_completion(self.net, generated, condition, label, *self.args)
        if self._key != key:
            out = refine_completion(self.net, generated, condition, label, *self.args)      # warm-up (lazy init)
            self._gen, self._cond = generated.clone(), condition.clone()
            self._label = None if label is None else label.clone()
            torch.cuda.current_stream(generated.device).synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._out = refine_completion(self.net, self._gen, self._cond, self._label, *self.args)
            self._graph, self._key = g, key
            return out
        self._gen.copy_(generated)
        self._cond.copy_(condition)
        if label is not None:
            self._label.copy_(label)
        self._graph.replay()
        return self._out.clone()


def gather_records(records, group=None, return_counts=False):
    """All ranks' (n_r, C) records concatenated in rank order -> (sum n_r, C) on every rank.
    Shards may have different lengths (last rank short): lengths are exchanged first and the
    payload is padded to the longest shard -- two tiny collectives for the whole job.
    return_counts: also return [n_0, ..., n_{W-1}] (how many records each rank contributed)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return (records, [int(records.shape[0])]) if return_counts else records
    world = dist.get_world_size(group)
    n = torch.tensor([records.shape[0]], dtype=torch.int64, device=records.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    width = records.shape[1]
    padded = torch.zeros((max(counts), width), dtype=records.dtype, device=records.device)
    padded[:records.shape[0]] = records
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    everything = torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)
    return (everything, counts) if return_counts else everything


def summarize(all_records):
    """Means exactly as generate_samples_distributed.py:84-95 computes them (concatenate, .mean())."""
    cd_t, cd_p, f1, emd = (all_records[:, i] for i in range(4))
    return {"avg_cd": float(cd_t.mean()), "avg_cd_p": float(cd_p.mean()), "avg_f1": float(f1.mean()),
            "avg_emd": float(emd.mean()), "num_samples": int(all_records.shape[0])}


def generate_and_evaluate(generate, dataset, num_shapes, batch_size, rank=0, world_size=1, scale=1.0,
                          compute_emd=True, group=None, device=None):
    """Run this rank's shard.  `dataset(lo, hi)` -> (condition, label, gt) for partial indices [lo, hi)
    (gt already expanded per partial view: gt_idx = index // 26, mvp_dataset.py:289) -- or, for a dataset that
    augments during generation (`augment_data_during_generation`, completion_eval.py:140-143), the 5-tuple
    (condition, label, gt, M_inv (b,3,3), translation (b,1,3)): generated clouds and gt are then mapped back with
    `matmul(x - translation, M_inv)` before the metrics, as completion_eval.py:203-211 does.
    `device`: where an EMPTY shard's (0,5) record tensor lives (a rank past the end of the data still takes part
    in the collective, and under RCCL every rank must hand over a tensor on its own GPU); defaults to the
    current CUDA device when the process group's backend is nccl, else the CPU.
    Returns (local generated clouds, ALL ranks' records in rank order, summary)."""
    first, last, _, _ = rank_shard(num_shapes, rank, world_size)
    clouds, recs = [], []
    for lo, hi in batches(first, last, batch_size):
        item = dataset(lo, hi)
        if len(item) not in (3, 5):
            raise ValueError("dataset(lo, hi) must return (condition, label, gt) or (condition, label, gt, M_inv, "
                             "translation); got %d items" % len(item))
        condition, label, gt = item[:3]
        M_inv, translation = item[3:] if len(item) == 5 else (None, None)
        g, r = evaluate_batch(generate, condition, label, gt, scale=scale, compute_emd=compute_emd, M_inv=M_inv,
                              translation=translation)
        clouds.append(g)
        recs.append(r)
    if recs:
        local = torch.cat(recs, 0)
    else:
        if device is None and dist.is_available() and dist.is_initialized() and \
                dist.get_backend(group) == "nccl" and torch.cuda.is_available():
            device = torch.device("cuda", torch.cuda.current_device())
        local = torch.zeros((0, 5), dtype=torch.float32, device=device)
    everything = gather_records(local, group=group)
    return (torch.cat(clouds, 0) if clouds else None), everything, summarize(everything)
