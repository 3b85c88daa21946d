"""Import the reference's Python layers IN THIS CONTAINER ONLY (fixture generation).

/root/reference never travels to the GPU box and is never read by tests, smoke()
or bench.py at run time.  This helper is used by tests/golden/make_golden.py to run
the reference's own Python (QueryAndGroup, group_knn, Mlp_plus_t_emb,
AttentionModule, PointNet2CloudCondition, sampling, VAR/STEP sampling, calc_cd,
emd.py, point_upsample) on the CPU, with the three native dependencies the image
lacks replaced by stand-ins backed by the CPU oracle (oracle/pdr_oracle.py):

    pointnet2_ops._ext            -> oracle ops (reference needs nvcc to build it)
    pytorch3d.ops.knn / structures-> oracle knn   (un-vendored third party)
    emd_cuda                      -> oracle approxmatch / matchcost

and `.cuda()` patched to identity.  What this pins is everything ABOVE the native
ops exactly as the reference composes it; the native arithmetic itself is pinned by
the oracle's own tests.
"""
import os
import sys
import types

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from oracle import pdr_oracle as O  # noqa: E402

REF = "/root/reference"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _np(t):
    return t.detach().cpu().contiguous().numpy()


def make_ext_standin():
    m = types.ModuleType("pointnet2_ops._ext")
    m.furthest_point_sampling = lambda pts, n: _t(O.furthest_point_sampling(_np(pts), n))
    m.gather_points = lambda pts, idx: _t(O.gather_points(_np(pts), _np(idx)))
    m.gather_points_grad = lambda g, idx, n: _t(O.gather_points_grad(_np(g), _np(idx), n))
    m.ball_query = lambda new_xyz, xyz, r, ns: tuple(_t(a) for a in O.ball_query(_np(new_xyz), _np(xyz), r, ns))
    m.group_points = lambda pts, idx: _t(O.group_points(_np(pts), _np(idx)))
    m.group_points_grad = lambda g, idx, n: _t(O.group_points_grad(_np(g), _np(idx), n))
    m.three_nn = lambda u, k: [_t(a) for a in O.three_nn(_np(u), _np(k))]
    m.three_interpolate = lambda p, i, w: _t(O.three_interpolate(_np(p), _np(i), _np(w)))
    m.three_interpolate_grad = lambda g, i, w, mm: _t(O.three_interpolate_grad(_np(g), _np(i), _np(w), mm))
    return m


class _KNN(tuple):
    dists = property(lambda s: s[0])
    idx = property(lambda s: s[1])
    knn = property(lambda s: s[2])


def _knn_points(p1, p2, lengths1=None, lengths2=None, K=1, version=-1, return_nn=False, return_sorted=True):
    d, i = O.knn(_np(p1), _np(p2), K)
    d, i = _t(d), _t(i)
    nn = _knn_gather(p2, i) if return_nn else None
    return _KNN((d, i, nn))


def _knn_gather(x, idx, lengths=None):
    B, M, C = x.shape
    _, N, K = idx.shape
    g = x[:, :, None].expand(-1, -1, K, -1).gather(1, idx.clamp(min=0)[:, :, :, None].expand(-1, -1, -1, C))
    return g


def install():
    """Pre-seed sys.modules and sys.path; idempotent."""
    if getattr(install, "_done", False):
        return
    sys.dont_write_bytecode = True
    sys.modules["pointnet2_ops._ext"] = make_ext_standin()
    p3 = types.ModuleType("pytorch3d")
    ops = types.ModuleType("pytorch3d.ops")
    knn = types.ModuleType("pytorch3d.ops.knn")
    knn.knn_points, knn.knn_gather = _knn_points, _knn_gather
    ops.knn = knn
    ops.knn_points, ops.knn_gather = _knn_points, _knn_gather
    st = types.ModuleType("pytorch3d.structures")
    pc = types.ModuleType("pytorch3d.structures.pointclouds")

    class Pointclouds:  # only isinstance() is ever applied to it on the dense path
        pass

    pc.Pointclouds = Pointclouds
    st.pointclouds = pc
    p3.ops, p3.structures = ops, st
    for name, mod in [("pytorch3d", p3), ("pytorch3d.ops", ops), ("pytorch3d.ops.knn", knn),
                      ("pytorch3d.structures", st), ("pytorch3d.structures.pointclouds", pc)]:
        sys.modules[name] = mod
    emd = types.ModuleType("emd_cuda")
    emd.approxmatch_forward = lambda a, b: _t(O.approxmatch(_np(a), _np(b)))
    emd.matchcost_forward = lambda a, b, m: _t(O.matchcost(_np(a), _np(b), _np(m)))
    emd.matchcost_backward = lambda g, a, b, m: [_t(x) for x in O.matchcost_grad(_np(g), _np(a), _np(b), _np(m))]
    sys.modules["emd_cuda"] = emd
    # data path (mvp_dataloader/mvp_dataset.py imports h5py; mvp_data_utils imports transforms3d): h5py.File as a
    # read-only view over libhdf5 through the product's ctypes binding -- what is being pinned is the READER's
    # semantics (splits, scaling, item dicts), the file format itself is pinned by the h5import-written fixture --
    # and the two transforms3d factors the augmentation uses, from their textbook definitions
    from point_diffusion_refinement_amd.pointnet2.mvp_dataloader import hdf5_io

    class _Dataset:
        def __init__(self, path, name):
            self._a = hdf5_io.read(path, name)
            self.shape, self.dtype = self._a.shape, self._a.dtype

        def __getitem__(self, key):
            return self._a[key]

        def __array__(self, dtype=None, copy=None):
            return self._a if dtype is None else self._a.astype(dtype)

        def __len__(self):
            return len(self._a)

    class _File:
        def __init__(self, path, mode='r'):
            assert mode == 'r', "stand-in h5py is read-only"
            self._path = path

        def __getitem__(self, name):
            return _Dataset(self._path, name)

        def close(self):
            pass

    h5 = types.ModuleType("h5py")
    h5.File = _File
    sys.modules["h5py"] = h5
    t3 = types.ModuleType("transforms3d")
    zooms, axangles = types.ModuleType("transforms3d.zooms"), types.ModuleType("transforms3d.axangles")

    def zfdir2mat(zfactor, direction=None):
        if direction is None:
            return np.diag([zfactor] * 3).astype(np.float64)
        n = np.asarray(direction, dtype=np.float64)
        n = n / np.linalg.norm(n)
        return np.eye(3) - (1 - zfactor) * np.outer(n, n)

    def axangle2mat(axis, angle):
        x, y, z = np.asarray(axis, dtype=np.float64) / np.linalg.norm(axis)
        c, s = np.cos(angle), np.sin(angle)
        C = 1 - c
        return np.array([[x * x * C + c, x * y * C - z * s, x * z * C + y * s],
                         [y * x * C + z * s, y * y * C + c, y * z * C - x * s],
                         [z * x * C - y * s, z * y * C + x * s, z * z * C + c]])
    zooms.zfdir2mat, axangles.axangle2mat = zfdir2mat, axangle2mat
    t3.zooms, t3.axangles = zooms, axangles
    for name, mod in [("transforms3d", t3), ("transforms3d.zooms", zooms), ("transforms3d.axangles", axangles)]:
        sys.modules[name] = mod
    # CPU-only container: .cuda() is the identity, tensors report is_cuda for emd.py's assert
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    for p in (REF, os.path.join(REF, "pointnet2"), os.path.join(REF, "pointnet2_ops_lib")):
        if p not in sys.path:
            sys.path.insert(0, p)
    install._done = True


class pretend_cuda:
    """emd.py:11 asserts `.is_cuda`; inside this context CPU tensors claim to be CUDA tensors."""

    def __enter__(self):
        torch.Tensor.is_cuda = property(lambda self: True)

    def __exit__(self, *a):
        del torch.Tensor.is_cuda   # fall back to the C-level descriptor of TensorBase


def load_config(name="config_standard_attention_real_3072_partial_points_rot_90_scale_1.2_translation_0.1.json"):
    import json
    install()
    from json_reader import restore_string_to_list_in_a_dict
    with open(os.path.join(REF, "pointnet2/exp_configs/mvp_configs", name)) as f:
        cfg = json.load(f)
    restore_string_to_list_in_a_dict(cfg)
    return cfg
