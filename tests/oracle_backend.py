"""TEST INFRASTRUCTURE: run the product's Python layers on CPU tensors by swapping the
native-op module (`pointnet2_ops._ext`, emd natives) for the CPU oracle.

The product itself has no CPU path (its `_ext` raises "CPU not supported", like the
reference).  Tests that exercise the host-side logic without a GPU -- module wiring,
state_dict compatibility, the sampler, sharding -- install this backend explicitly.
"""
import contextlib

import numpy as np
import torch

from oracle import pdr_oracle as O
from point_diffusion_refinement_amd.pointnet2 import emd as _emd
from point_diffusion_refinement_amd.pointnet2_ops import _ext


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _n(t):
    return t.detach().cpu().contiguous().numpy()


def _knn_points(p1, p2, K, return_nn=False):
    d, i = O.knn(_n(p1), _n(p2), K)
    d, i = _t(d), _t(i)
    nn = None
    if return_nn:
        B, n2, _ = p2.shape
        nn = p2.gather(1, i.clamp(min=0).reshape(B, -1, 1).expand(-1, -1, 3)).view(B, -1, K, 3)
    return d, i, nn


def _chamfer_nn(x, y):
    dx, ix = O.knn(_n(x), _n(y), 1)
    dy, iy = O.knn(_n(y), _n(x), 1)
    return _t(dx[..., 0]), _t(ix[..., 0]), _t(dy[..., 0]), _t(iy[..., 0])


ORACLE_EXT = {
    "furthest_point_sampling": lambda pts, n: _t(O.furthest_point_sampling(_n(pts), n)),
    "gather_points": lambda pts, idx: _t(O.gather_points(_n(pts), _n(idx))),
    "gather_points_grad": lambda g, idx, n: _t(O.gather_points_grad(_n(g), _n(idx), n)),
    "ball_query": lambda q, xyz, r, ns: tuple(_t(a) for a in O.ball_query(_n(q), _n(xyz), r, ns)),
    "group_points": lambda pts, idx: _t(O.group_points(_n(pts), _n(idx))),
    "group_points_grad": lambda g, idx, n: _t(O.group_points_grad(_n(g), _n(idx), n)),
    "three_nn": lambda u, k: [_t(a) for a in O.three_nn(_n(u), _n(k))],
    "three_interpolate": lambda p, i, w: _t(O.three_interpolate(_n(p), _n(i), _n(w))),
    "three_interpolate_grad": lambda g, i, w, m: _t(O.three_interpolate_grad(_n(g), _n(i), _n(w), m)),
    "knn_points": _knn_points,
    "chamfer_nn": _chamfer_nn,
}


@contextlib.contextmanager
def oracle_ops():
    """Context manager: product Python layers run over the oracle on CPU tensors."""
    saved = {k: getattr(_ext, k) for k in ORACLE_EXT}
    saved_emd = (_emd.emd_cost_fused, _emd.approxmatch_forward, _emd.matchcost_forward)
    for k, v in ORACLE_EXT.items():
        setattr(_ext, k, v)
    _emd.approxmatch_forward = lambda a, b: _t(O.approxmatch(_n(a), _n(b)))
    _emd.matchcost_forward = lambda a, b, m: _t(O.matchcost(_n(a), _n(b), _n(m)))
    _emd.emd_cost_fused = lambda a, b: _t(O.matchcost(_n(a), _n(b), O.approxmatch(_n(a), _n(b))))
    try:
        yield
    finally:
        for k, v in saved.items():
            setattr(_ext, k, v)
        _emd.emd_cost_fused, _emd.approxmatch_forward, _emd.matchcost_forward = saved_emd
