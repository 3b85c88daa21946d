"""Data path of SURVEY 8(f)3 on CPU: HDF5 / npz containers, the MVP shard reader (reference
mvp_dataset.py:16-328 semantics), mirror preprocessing feeding it, per-rank result writers and
`gather_generated_results` (generate_samples_distributed.py:26-97)."""
import os
import pickle

import numpy as np
import pytest
import torch

from point_diffusion_refinement_amd.pointnet2 import generation as G
from point_diffusion_refinement_amd.pointnet2.mvp_dataloader import hdf5_io, results, shard_io
from point_diffusion_refinement_amd.pointnet2.mvp_dataloader.generate_mirrored_partial import build_mirrored_partials
from point_diffusion_refinement_amd.pointnet2.mvp_dataloader.mvp_dataset import ShapeNetH5
from tests.oracle_backend import oracle_ops

EXT = ".h5" if hdf5_io.available() else ".npz"


@pytest.mark.skipif(not hdf5_io.available(), reason="libhdf5 not present")
def test_hdf5_round_trip_and_format(tmp_path):
    rng = np.random.default_rng(0)
    arrays = {"data": rng.random((5, 7, 3)).astype(np.float32), "labels": np.arange(5, dtype=np.int64),
              "bytes": np.arange(6, dtype=np.uint8).reshape(2, 3), "d": rng.random((4,)), "i32": np.array([[-3, 9]], np.int32),
              "empty": np.zeros((0, 3), np.float32)}
    path = str(tmp_path / "x.h5")
    assert shard_io.save_arrays(path, arrays) == path
    assert open(path, "rb").read(8) == b"\x89HDF\r\n\x1a\n"                  # a real HDF5 file (h5py-readable)
    for k, v in arrays.items():
        r = shard_io.load_array(path, k)
        assert r.dtype == v.dtype and r.shape == v.shape and np.array_equal(r, v), k
    with pytest.raises(KeyError):
        shard_io.load_array(path, "missing")
    with pytest.raises(FileNotFoundError):
        shard_io.load_array(str(tmp_path / "nope.h5"), "data")


@pytest.mark.skipif(not hdf5_io.available(), reason="libhdf5 not present")
def test_reads_a_file_written_by_the_hdf_groups_h5import(tmp_path):
    """tests/golden/h5/h5import_written.h5 was produced by `h5import` (HDF Group command-line tool, make_fixture.sh),
    not by this repo: contiguous float32 / int64 datasets and a chunked, gzip-compressed float64 dataset in a group.
    And the other direction where the tools exist: a file written here is listed and dumped correctly by h5ls / h5dump."""
    import shutil
    import subprocess
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")
    want = np.load(os.path.join(d, "expect.npz"))
    path = os.path.join(d, "h5import_written.h5")
    for name, key in (("incomplete_pcds", "incomplete_pcds"), ("labels", "labels"), ("group/data", "data")):
        got = hdf5_io.read(path, name)
        assert got.dtype == want[key].dtype and np.array_equal(got, want[key]), name
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):
        mine = str(tmp_path / "mine.h5")
        a = (np.arange(24, dtype=np.float32) / 8 - 1).reshape(2, 4, 3)
        hdf5_io.write(mine, {"complete_pcds": a, "labels": np.array([7, -1], np.int64)})
        text = subprocess.run([h5dump, "-d", "/complete_pcds", "-y", "-w", "0", mine], capture_output=True, text=True).stdout
        assert "H5T_IEEE_F32LE" in text and "( 2, 4, 3 )" in text
        body = text[text.index("DATA {") + 6:text.rindex("}")]
        vals = np.array([float(v) for v in body.replace("}", " ").replace("\n", " ").split(",") if v.strip()], np.float32)
        assert np.array_equal(vals, a.reshape(-1))


def test_npz_fallback_is_transparent(tmp_path):
    a = np.arange(24, dtype=np.float32).reshape(2, 4, 3)
    shard_io.save_arrays(str(tmp_path / "y.npz"), {"data": a})
    # the reader is asked for the reference's file name and finds the .npz sibling
    assert np.array_equal(shard_io.load_array(str(tmp_path / "y.h5"), "data"), a)


def _make_mvp(root, G_norm=5, G_novel=2, n=48, npoints=64, split="test"):
    rng = np.random.default_rng(7)
    def clouds(g, pts):
        return (rng.random((g, pts, 3)).astype(np.float32) - 0.5)
    gt, ngt = clouds(G_norm, npoints), clouds(G_novel, npoints)
    # 26 partial views per shape: a random subset of the complete cloud
    def partials(g):
        return np.stack([c[rng.permutation(npoints)[:n]] for c in g for _ in range(26)])
    inp, ninp = partials(gt), partials(ngt)
    labels = np.repeat(np.arange(G_norm) % 8, 26).astype(np.int64)
    nlabels = np.repeat(8 + np.arange(G_novel), 26).astype(np.int64)
    shard_io.save_arrays(os.path.join(root, "mvp_%s_input%s" % (split, EXT)),
                         {"incomplete_pcds": inp, "labels": labels, "novel_incomplete_pcds": ninp, "novel_labels": nlabels})
    shard_io.save_arrays(os.path.join(root, "mvp_%s_gt_%dpts%s" % (split, npoints, EXT)),
                         {"complete_pcds": gt, "novel_complete_pcds": ngt})
    return inp, ninp, gt, ngt, labels, nlabels


def test_dataset_matches_reference_semantics(tmp_path):
    root = str(tmp_path)
    inp, ninp, gt, ngt, labels, nlabels = _make_mvp(root)
    ds = ShapeNetH5(root, train=False, npoints=64, scale=1.2)
    assert len(ds) == 7 * 26
    np.testing.assert_allclose(ds.input_data, np.concatenate([inp, ninp]) * 2 * 1.2, rtol=1e-6)
    item = ds[26 * 5 + 3]                                                     # a novel-category view
    assert item["label"] == 8 and torch.equal(item["complete"], torch.from_numpy(ngt[0] * 2 * np.float32(1.2)))
    assert item["partial"].shape == (48, 3)
    only = ShapeNetH5(root, train=False, npoints=64, novel_input_only=True)
    assert len(only) == 2 * 26 and only.labels.min() == 8
    # rank split == generation.rank_shard == the reference's ceil(G/W) rule; ranks concatenate to the whole
    parts = [ShapeNetH5(root, train=False, npoints=64, scale=1.2, rank=r, world_size=3,
                        append_samples_to_last_rank=False) for r in range(3)]
    for r, p in enumerate(parts):
        lo, hi, s0, s1 = G.rank_shard(7, r, 3)
        assert len(p) == hi - lo and p.gt_data.shape[0] == s1 - s0
        assert np.array_equal(p.input_data, ds.input_data[lo:hi]) and np.array_equal(p.labels, ds.labels[lo:hi])
    assert sum(len(p) for p in parts) == len(ds) and len(parts[2]) == 26     # last rank short: ceil(7/3) = 3
    cond, label, gtb = parts[1].batch(20, 40)
    assert cond.shape == (20, 48, 3) and label.dtype == torch.int64
    assert torch.equal(gtb[0], torch.from_numpy(parts[1].gt_data[20 // 26])) and \
        torch.equal(gtb[-1], torch.from_numpy(parts[1].gt_data[39 // 26]))


def test_mirrored_partials_feed_the_dataset(tmp_path):
    root = str(tmp_path)
    inp, ninp, *_ = _make_mvp(root)
    with oracle_ops(), torch.no_grad():                                       # FPS / gather on the CPU oracle
        paths = build_mirrored_partials(root, train=False, batch_size=64, device="cpu", num_points=(40, 64), npoints=64)
    assert [os.path.basename(p).split("concat_")[1].split("pts")[0] for p in paths] == ["96", "40", "64"]
    ds = ShapeNetH5(root, train=False, npoints=64, scale=1.5, use_mirrored_partial_input=True, number_partial_points=64)
    assert ds.input_data.shape == (7 * 26, 64, 4)
    assert set(np.unique(ds.input_data[:, :, 3])) == {-1.0, 1.0}              # tag channel is not rescaled
    # every selected row is a point of the cloud or of its z-mirror, rescaled by 2 * scale
    raw = np.concatenate([inp, ninp])[5]
    both = np.concatenate([raw, raw * np.array([1, 1, -1], np.float32)]) * 3.0
    sel = ds.input_data[5][:, :3]
    d = np.abs(sel[:, None] - both[None]).sum(-1).min(1)
    assert d.max() < 1e-6
    full = shard_io.load_array(paths[0], "data")
    assert full.shape == (7 * 26, 96, 4) and np.array_equal(full[:, :48, 3], np.ones((7 * 26, 48), np.float32))


def test_rank_results_gather_equals_concatenation(tmp_path):
    root = str(tmp_path)
    rng = np.random.default_rng(3)
    gens = [rng.random((n, 32, 3)).astype(np.float32) for n in (6, 6, 2)]
    recs = [rng.random((n, 5)).astype(np.float32) for n in (6, 6, 2)]
    for r in range(3):
        recs[r][:, 4] = r
        results.save_rank_results(root, r, gens[r], recs[r], iteration=123, dataset="mvp")
    out = results.gather_generated_results(root, 3, remove_original_files=True)
    allrec = np.concatenate(recs)
    assert np.array_equal(out["cd_distance"], allrec[:, 0]) and np.array_equal(out["emd_distance"], allrec[:, 3])
    assert np.array_equal(out["f1"], allrec[:, 2]) and np.array_equal(out["meta"], allrec[:, 4].astype(np.int64))
    assert out["iter"] == 123 and np.isclose(out["avg_cd"], allrec[:, 0].mean())
    # the same table generation.summarize works on (in-memory all-gather form)
    s = G.summarize(torch.from_numpy(allrec))
    assert np.isclose(s["avg_cd"], out["avg_cd"]) and np.isclose(s["avg_emd"], out["avg_emd"])
    data = shard_io.load_array(os.path.join(root, "mvp_generated_data_32pts.h5"), "data")
    assert np.array_equal(data, np.concatenate(gens))
    with open(os.path.join(root, "mvp_eval_result.pkl"), "rb") as h:
        assert set(pickle.load(h)) == {"meta", "cd_distance", "emd_distance", "f1", "avg_cd", "avg_emd", "iter"}
    assert os.listdir(os.path.join(root, "rank_0")) == []                     # originals removed
    assert "CD loss" in open(os.path.join(root, "gathered_generation.log")).read()


# ------------------------------------------------------------------ pinned against the REFERENCE reader
def test_shapenet_h5_equals_the_reference_reader(tmp_path):
    """tests/golden/dataset.npz was produced by the reference's own mvp_dataset.ShapeNetH5 (imported in the build
    container over stand-ins for h5py / transforms3d, tests/golden/make_golden.py dataset()) on the tiny synthetic MVP
    directory rebuilt here: rank splits incl. the short last rank, the mirrored 4-channel input, the topped-up last
    training rank (same `random.sample` draws), random subsampling, novel-only, scale != 1, and an augmented item
    with M_inv / translation -- arrays and items bit for bit."""
    import random
    from tests.golden import dataset_inputs as DI
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset.npz"))
    src = {k[4:]: g[k] for k in g.files if k.startswith("src_")}
    for k, v in DI.source_arrays().items():                              # the committed inputs are the seeded ones
        assert np.array_equal(src[k], v), k
    ext = EXT
    DI.write_directory(str(tmp_path), src, lambda path, arrays: shard_io.save_arrays(path[:-3] + ext, arrays))
    for name, kw, seed in DI.CASES:
        random.seed(seed), np.random.seed(seed)
        ds = ShapeNetH5(str(tmp_path), **kw)
        assert len(ds) == int(g[name + "_len"]), name
        for attr, key in (("input_data", "_input"), ("gt_data", "_gt"), ("labels", "_labels")):
            got, want = getattr(ds, attr), g[name + key]
            assert got.shape == want.shape and np.array_equal(got, want), (name, attr)
            assert got.dtype == want.dtype or attr == "labels", (name, attr, got.dtype, want.dtype)
        if name + "_p2c" in g.files:
            assert np.array_equal(ds.partial_to_complete_index, g[name + "_p2c"])
        for i in DI.item_indices(len(ds)):
            random.seed(seed + i), np.random.seed(seed + i)
            item = ds[i]
            keys = sorted(k.split("_item%d_" % i)[1] for k in g.files if k.startswith("%s_item%d_" % (name, i)))
            assert sorted(item.keys()) == keys, (name, i, sorted(item.keys()), keys)
            for k in keys:
                want = g["%s_item%d_%s" % (name, i, k)]
                got = item[k].numpy() if torch.is_tensor(item[k]) else np.asarray(item[k])
                assert got.shape == want.shape and np.array_equal(got, want), (name, i, k)
    assert (g["test_w2_r1_len"], g["test_w2_r0_len"]) == (3 * 26, 4 * 26)      # the last rank IS short
    assert "test_augmented_item0_M_inv" in g.files and g["train_w2_r1_append_len"] == 4 * 26


def test_deaugmentation_of_generated_clouds_matches_the_reference_harness(monkeypatch):
    """generation.evaluate_batch(M_inv=, translation=): completion_eval.py:203-205 `matmul(x - translation, M_inv)`
    on generated clouds AND ground truth before the /2/scale and the metrics (golden from the reference expression)."""
    from tests.golden import dataset_inputs as DI
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset.npz"))
    gen, gt, M_inv, tr = DI.deaugment_inputs()
    seen = {}

    def fake_cd(generated, gtt, calc_f1=True, f1_threshold=1e-4):
        seen["generated"], seen["gt"] = generated.clone(), gtt.clone()
        z = torch.zeros(generated.shape[0])
        return z, z, z
    from point_diffusion_refinement_amd.pointnet2 import chamfer_loss_new
    monkeypatch.setattr(chamfer_loss_new, "calc_cd", fake_cd)
    out, rec = G.evaluate_batch(lambda c, l: gen, None, torch.zeros(3, dtype=torch.long), gt, scale=0.5, compute_emd=False,
                                M_inv=M_inv, translation=tr)
    assert np.array_equal(out.numpy(), g["deaug_generated"] / 2 / 0.5)
    assert np.array_equal(seen["gt"].numpy(), g["deaug_gt"] / 2 / 0.5) and rec.shape == (3, 5)
