"""numpy front-end of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  It loads oracle/libpdr_oracle.so (built by oracle/Makefile from
oracle/pdr_oracle.c) through ctypes; arrays are numpy, C-contiguous.

Function names and argument orders follow the reference's pybind surface
(pointnet2_ops/_ext-src/src/bindings.cpp:6-19, PytorchEMD/cuda/emd.cpp:23-27)
so the stand-in modules in tests/golden/make_golden.py are one-liners.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpdr_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "pdr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libpdr_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _chk(rc, name):
    if rc != 0:
        raise RuntimeError("oracle %s failed with code %d" % (name, rc))


def opt_n_threads(n):
    return int(lib().pdr_oracle_opt_n_threads(ctypes.c_int(int(n))))


def furthest_point_sampling(xyz, nsamples):
    xyz = _f32(xyz)
    B, N, _ = xyz.shape
    out = np.zeros((B, nsamples), dtype=np.int32)
    _chk(lib().pdr_oracle_furthest_point_sampling(_p(xyz), B, N, int(nsamples), _p(out)), "fps")
    return out


def gather_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    B, C, N = points.shape
    m = idx.shape[1]
    out = np.zeros((B, C, m), dtype=np.float32)
    _chk(lib().pdr_oracle_gather_points(_p(points), _p(idx), B, C, N, m, _p(out)), "gather_points")
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    B, C, m = grad_out.shape
    out = np.zeros((B, C, n), dtype=np.float32)
    _chk(lib().pdr_oracle_gather_points_grad(_p(grad_out), _p(idx), B, C, int(n), m, _p(out)), "gather_points_grad")
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    new_xyz, xyz = _f32(new_xyz), _f32(xyz)
    B, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = np.zeros((B, m, nsample), dtype=np.int32)
    cnt = np.zeros((B, m), dtype=np.int32)
    _chk(lib().pdr_oracle_ball_query(_p(new_xyz), _p(xyz), B, n, m, ctypes.c_float(radius),
                                     int(nsample), _p(idx), _p(cnt)), "ball_query")
    return idx, cnt


def group_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    B, C, N = points.shape
    _, npnt, ns = idx.shape
    out = np.zeros((B, C, npnt, ns), dtype=np.float32)
    _chk(lib().pdr_oracle_group_points(_p(points), _p(idx), B, C, N, npnt, ns, _p(out)), "group_points")
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    B, C, npnt, ns = grad_out.shape
    out = np.zeros((B, C, n), dtype=np.float32)
    _chk(lib().pdr_oracle_group_points_grad(_p(grad_out), _p(idx), B, C, int(n), npnt, ns, _p(out)), "group_points_grad")
    return out


def three_nn(unknown, known):
    unknown, known = _f32(unknown), _f32(known)
    B, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.zeros((B, n, 3), dtype=np.float32)
    idx = np.zeros((B, n, 3), dtype=np.int32)
    _chk(lib().pdr_oracle_three_nn(_p(unknown), _p(known), B, n, m, _p(d2), _p(idx)), "three_nn")
    return d2, idx


def three_interpolate(points, idx, weight):
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    B, C, m = points.shape
    n = idx.shape[1]
    out = np.zeros((B, C, n), dtype=np.float32)
    _chk(lib().pdr_oracle_three_interpolate(_p(points), _p(idx), _p(weight), B, C, m, n, _p(out)), "three_interpolate")
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, idx, weight = _f32(grad_out), _i32(idx), _f32(weight)
    B, C, n = grad_out.shape
    out = np.zeros((B, C, m), dtype=np.float32)
    _chk(lib().pdr_oracle_three_interpolate_grad(_p(grad_out), _p(idx), _p(weight), B, C, n, int(m), _p(out)),
         "three_interpolate_grad")
    return out


def knn(x, y, K):
    """pytorch3d.ops.knn_points contract: (dists (B,n1,K) f32 squared, idx (B,n1,K) i64)."""
    x, y = _f32(x), _f32(y)
    B, n1, _ = x.shape
    n2 = y.shape[1]
    d = np.zeros((B, n1, K), dtype=np.float32)
    idx = np.zeros((B, n1, K), dtype=np.int64)
    _chk(lib().pdr_oracle_knn(_p(x), _p(y), B, n1, n2, int(K), _p(d), _p(idx)), "knn")
    return d, idx


def knn_mink(x, y, K):
    """The same op through the restatement of pytorch3d's published MinK (replace the current maximum on a strictly
    smaller key, stable bubble sort): the comparison object of tests/test_oracle.py, not a contract of its own."""
    x, y = _f32(x), _f32(y)
    B, n1, _ = x.shape
    n2 = y.shape[1]
    d = np.zeros((B, n1, K), dtype=np.float32)
    idx = np.zeros((B, n1, K), dtype=np.int64)
    _chk(lib().pdr_oracle_knn_mink(_p(x), _p(y), B, n1, n2, int(K), _p(d), _p(idx)), "knn_mink")
    return d, idx


def knn_grad(x, y, idx, grad_dists):
    """Backward of knn(): (grad_x (B,n1,3), grad_y (B,n2,3))."""
    x, y, grad_dists = _f32(x), _f32(y), _f32(grad_dists)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    B, n1, _ = x.shape
    n2 = y.shape[1]
    K = idx.shape[2]
    gx = np.zeros((B, n1, 3), dtype=np.float32)
    gy = np.zeros((B, n2, 3), dtype=np.float32)
    _chk(lib().pdr_oracle_knn_grad(_p(x), _p(y), _p(idx), _p(grad_dists), B, n1, n2, int(K), _p(gx), _p(gy)),
         "knn_grad")
    return gx, gy


def approxmatch(xyz1, xyz2):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    match = np.zeros((B, m, n), dtype=np.float32)
    _chk(lib().pdr_oracle_approxmatch(_p(xyz1), _p(xyz2), B, n, m, _p(match)), "approxmatch")
    return match


def matchcost(xyz1, xyz2, match):
    xyz1, xyz2, match = _f32(xyz1), _f32(xyz2), _f32(match)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    cost = np.zeros((B,), dtype=np.float32)
    _chk(lib().pdr_oracle_matchcost(_p(xyz1), _p(xyz2), _p(match), B, n, m, _p(cost)), "matchcost")
    return cost


def matchcost_grad(grad_cost, xyz1, xyz2, match):
    grad_cost, xyz1, xyz2, match = _f32(grad_cost), _f32(xyz1), _f32(xyz2), _f32(match)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = np.zeros((B, n, 3), dtype=np.float32)
    g2 = np.zeros((B, m, 3), dtype=np.float32)
    _chk(lib().pdr_oracle_matchcost_grad(_p(grad_cost), _p(xyz1), _p(xyz2), _p(match), B, n, m, _p(g1), _p(g2)),
         "matchcost_grad")
    return g1, g2


def emd(xyz1, xyz2):
    """pointnet2/emd.py:12-16: cost / max(n, m)."""
    match = approxmatch(xyz1, xyz2)
    cost = matchcost(xyz1, xyz2, match)
    return cost / max(xyz1.shape[1], xyz2.shape[1])


def chamfer(x, y):
    """chamfer_loss_new.py:149-153 via knn K=1: (dist_x (B,P1), idx_x, dist_y (B,P2), idx_y)."""
    dx, ix = knn(x, y, 1)
    dy, iy = knn(y, x, 1)
    return dx[..., 0], ix[..., 0], dy[..., 0], iy[..., 0]
