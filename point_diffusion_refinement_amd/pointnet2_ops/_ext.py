"""Stand-in for the reference's compiled module `pointnet2_ops._ext`.

Same nine callables with the same argument order and return types as the pybind
module defined in _ext-src/src/bindings.cpp:6-19, plus `knn_points` (the
pytorch3d.ops.knn entry the reference's Python layer calls).  Each function
validates like the reference's AT_ASSERT macros (_ext-src/include/utils.h:5-25:
contiguous, dtype, device) -- raising RuntimeError instead of aborting --,
allocates the outputs as torch tensors and forwards raw device pointers plus the
CURRENT stream to libpdr_hip.so.  CPU tensors are rejected exactly like the
reference ("CPU not supported", sampling.cpp:34): there is no CPU fallback.
"""
import torch

from .. import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _req(t, name, dtype):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s: CPU not supported" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be a %s tensor" % (name, "float" if dtype == torch.float32 else "int"))


def _req_xyz(t, name):
    """(B, N, 3) coordinate clouds: the kernels read three floats per point."""
    if t.dim() != 3 or t.shape[2] != 3:
        raise RuntimeError("%s must have shape (B, N, 3), got %s" % (name, tuple(t.shape)))


def _same_device(*ts):
    d = ts[0].device
    for t in ts[1:]:
        if t.device != d:
            raise RuntimeError("all tensors must live on the same device")


def furthest_point_sampling(points, nsamples):
    """(B,N,3) f32 -> (B,nsamples) i32   [sampling.cpp:66-87]"""
    _req(points, "points", torch.float32)
    _req_xyz(points, "points")
    B, N, _ = points.shape
    lib = _lib.load()
    out = torch.empty((B, nsamples), dtype=torch.int32, device=points.device)
    ws = lib.pdr_fps_workspace_bytes(B, N)
    temp = torch.empty((ws // 4,), dtype=torch.float32, device=points.device) if ws else None
    with torch.cuda.device(points.device):
        _lib.check(lib.pdr_furthest_point_sampling(points.data_ptr(), B, N, int(nsamples),
                                                   temp.data_ptr() if temp is not None else None,
                                                   out.data_ptr(), _stream()), "furthest_point_sampling")
    return out


def gather_points(points, idx):
    """(B,C,N) f32, (B,m) i32 -> (B,C,m)   [sampling.cpp:15-38]"""
    _req(points, "points", torch.float32)
    _req(idx, "idx", torch.int32)
    _same_device(points, idx)
    B, C, N = points.shape
    m = idx.shape[1]
    out = torch.empty((B, C, m), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _lib.check(_lib.load().pdr_gather_points(points.data_ptr(), idx.data_ptr(), B, C, N, m,
                                                 out.data_ptr(), _stream()), "gather_points")
    return out


def gather_points_grad(grad_out, idx, n):
    """(B,C,m) f32, (B,m) i32, n -> (B,C,n)   [sampling.cpp:40-64]"""
    _req(grad_out, "grad_out", torch.float32)
    _req(idx, "idx", torch.int32)
    _same_device(grad_out, idx)
    B, C, m = grad_out.shape
    out = torch.empty((B, C, n), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _lib.check(_lib.load().pdr_gather_points_grad(grad_out.data_ptr(), idx.data_ptr(), B, C, int(n), m,
                                                      out.data_ptr(), _stream()), "gather_points_grad")
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """(B,m,3), (B,n,3), r, ns -> (idx (B,m,ns) i32, counts (B,m) i32)   [ball_query.cpp:10-38]"""
    _req(new_xyz, "new_xyz", torch.float32)
    _req(xyz, "xyz", torch.float32)
    _req_xyz(new_xyz, "new_xyz")
    _req_xyz(xyz, "xyz")
    _same_device(new_xyz, xyz)
    if new_xyz.shape[0] != xyz.shape[0]:
        raise RuntimeError("new_xyz and xyz must have the same batch size")
    B, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = torch.empty((B, m, nsample), dtype=torch.int32, device=xyz.device)
    counts = torch.empty((B, m), dtype=torch.int32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _lib.check(_lib.load().pdr_ball_query(new_xyz.data_ptr(), xyz.data_ptr(), B, n, m, float(radius),
                                              int(nsample), idx.data_ptr(), counts.data_ptr(), _stream()),
                   "ball_query")
    return idx, counts


def group_points(points, idx):
    """(B,C,N) f32, (B,np,ns) i32 -> (B,C,np,ns)   [group_points.cpp:12-36]"""
    _req(points, "points", torch.float32)
    _req(idx, "idx", torch.int32)
    _same_device(points, idx)
    B, C, N = points.shape
    _, npnt, ns = idx.shape
    out = torch.empty((B, C, npnt, ns), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _lib.check(_lib.load().pdr_group_points(points.data_ptr(), idx.data_ptr(), B, C, N, npnt, ns,
                                                out.data_ptr(), _stream()), "group_points")
    return out


def group_points_grad(grad_out, idx, n):
    """(B,C,np,ns) f32, (B,np,ns) i32, n -> (B,C,n)   [group_points.cpp:38-64]"""
    _req(grad_out, "grad_out", torch.float32)
    _req(idx, "idx", torch.int32)
    _same_device(grad_out, idx)
    B, C, npnt, ns = grad_out.shape
    out = torch.empty((B, C, n), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _lib.check(_lib.load().pdr_group_points_grad(grad_out.data_ptr(), idx.data_ptr(), B, C, int(n), npnt, ns,
                                                     out.data_ptr(), _stream()), "group_points_grad")
    return out


def three_nn(unknowns, knows):
    """(B,n,3), (B,m,3) -> [dist2 (B,n,3) f32 SQUARED, idx (B,n,3) i32]   [interpolate.cpp:14-40]"""
    _req(unknowns, "unknowns", torch.float32)
    _req(knows, "knows", torch.float32)
    _req_xyz(unknowns, "unknowns")
    _req_xyz(knows, "knows")
    _same_device(unknowns, knows)
    B, n, _ = unknowns.shape
    m = knows.shape[1]
    dist2 = torch.empty((B, n, 3), dtype=torch.float32, device=unknowns.device)
    idx = torch.empty((B, n, 3), dtype=torch.int32, device=unknowns.device)
    with torch.cuda.device(unknowns.device):
        _lib.check(_lib.load().pdr_three_nn(unknowns.data_ptr(), knows.data_ptr(), B, n, m, dist2.data_ptr(),
                                            idx.data_ptr(), _stream()), "three_nn")
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    """(B,C,m) f32, (B,n,3) i32, (B,n,3) f32 -> (B,C,n)   [interpolate.cpp:42-70]"""
    _req(points, "points", torch.float32)
    _req(idx, "idx", torch.int32)
    _req(weight, "weight", torch.float32)
    _same_device(points, idx, weight)
    B, C, m = points.shape
    n = idx.shape[1]
    out = torch.empty((B, C, n), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _lib.check(_lib.load().pdr_three_interpolate(points.data_ptr(), idx.data_ptr(), weight.data_ptr(), B, C, m,
                                                     n, out.data_ptr(), _stream()), "three_interpolate")
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """(B,C,n) f32, (B,n,3) i32, (B,n,3) f32, m -> (B,C,m)   [interpolate.cpp:72-100]"""
    _req(grad_out, "grad_out", torch.float32)
    _req(idx, "idx", torch.int32)
    _req(weight, "weight", torch.float32)
    _same_device(grad_out, idx, weight)
    B, C, n = grad_out.shape
    out = torch.empty((B, C, m), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _lib.check(_lib.load().pdr_three_interpolate_grad(grad_out.data_ptr(), idx.data_ptr(), weight.data_ptr(), B,
                                                          C, n, int(m), out.data_ptr(), _stream()),
                   "three_interpolate_grad")
    return out


def _knn_forward(p1, p2, K, return_nn):
    B, n1, _ = p1.shape
    n2 = p2.shape[1]
    dists = torch.empty((B, n1, K), dtype=torch.float32, device=p1.device)
    idx = torch.empty((B, n1, K), dtype=torch.int64, device=p1.device)
    nn = torch.empty((B, n1, K, 3), dtype=torch.float32, device=p1.device) if return_nn else None
    with torch.cuda.device(p1.device):
        _lib.check(_lib.load().pdr_knn_points(p1.data_ptr(), p2.data_ptr(), B, n1, n2, int(K), dists.data_ptr(),
                                              idx.data_ptr(), nn.data_ptr() if return_nn else None, _stream()),
                   "knn_points")
    return dists, idx, nn


def chamfer_nn(x, y):
    """Both K = 1 searches of a Chamfer evaluation in one launch:
    (B,n1,3), (B,n2,3) -> (dist_xy (B,n1), idx_xy (B,n1) i64, dist_yx (B,n2), idx_yx (B,n2) i64); not differentiable
    (chamfer_distance uses knn_points when a gradient is needed)."""
    _req(x, "x", torch.float32)
    _req(y, "y", torch.float32)
    _same_device(x, y)
    if x.shape[2] != 3 or y.shape[2] != 3 or x.shape[0] != y.shape[0]:
        raise RuntimeError("chamfer_nn: (B,n,3) clouds with equal batch size")
    B, n1, _ = x.shape
    n2 = y.shape[1]
    dx = torch.empty((B, n1), dtype=torch.float32, device=x.device)
    dy = torch.empty((B, n2), dtype=torch.float32, device=x.device)
    ix = torch.empty((B, n1), dtype=torch.int64, device=x.device)
    iy = torch.empty((B, n2), dtype=torch.int64, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().pdr_chamfer_nn(x.data_ptr(), y.data_ptr(), B, n1, n2, dx.data_ptr(), ix.data_ptr(),
                                              dy.data_ptr(), iy.data_ptr(), _stream()), "chamfer_nn")
    return dx, ix, dy, iy


def knn_group(x, y, K):
    """knn_points for group_knn in the fused network: (dists (B,n1,K) f32, idx (B,n1,K) i32, weights (B,n1,K) f32)
    with weights = normalised 1 / (d2 + 1e-8) (pointnet2_utils.py:500-503)."""
    _req(x, "x", torch.float32)
    _req(y, "y", torch.float32)
    _same_device(x, y)
    if x.dim() != 3 or y.dim() != 3 or x.shape[2] != 3 or y.shape[2] != 3 or x.shape[0] != y.shape[0]:
        raise RuntimeError("knn_group: x (B,n1,3) and y (B,n2,3) with equal batch size required, got %s and %s"
                           % (tuple(x.shape), tuple(y.shape)))
    B, n1, _ = x.shape
    n2 = y.shape[1]
    K = int(K)
    if K > min(n2, 16):
        # outside pdr_knn_group's contract (no padding slots, K <= 16): knn_points pads like pytorch3d (idx -1,
        # distance 0); weights as group_knn computes them from the squared distances
        d, i, _ = knn_points(x, y, K)
        w = 1.0 / (d + 1e-8)
        w = w / w.sum(dim=2, keepdim=True)
        return d, i.clamp(min=0).to(torch.int32), w
    d = torch.empty((B, n1, K), dtype=torch.float32, device=x.device)
    w = torch.empty((B, n1, K), dtype=torch.float32, device=x.device)
    i = torch.empty((B, n1, K), dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().pdr_knn_group(x.data_ptr(), y.data_ptr(), B, n1, n2, int(K), d.data_ptr(),
                                             i.data_ptr(), w.data_ptr(), _stream()), "knn_group")
    return d, i, w


class _KnnDists(torch.autograd.Function):
    """Differentiable squared distances of knn_points (pytorch3d `_knn_points` backward, norm 2)."""

    @staticmethod
    def forward(ctx, p1, p2, K):
        dists, idx, _ = _knn_forward(p1, p2, K, False)
        ctx.save_for_backward(p1, p2, idx)
        ctx.mark_non_differentiable(idx)
        return dists, idx

    @staticmethod
    def backward(ctx, grad_dists, _grad_idx):
        p1, p2, idx = ctx.saved_tensors
        B, n1, _ = p1.shape
        n2, K = p2.shape[1], idx.shape[2]
        g = grad_dists.contiguous().float()
        g1 = torch.empty_like(p1)
        g2 = torch.empty_like(p2)
        with torch.cuda.device(p1.device):
            _lib.check(_lib.load().pdr_knn_points_grad(p1.data_ptr(), p2.data_ptr(), idx.data_ptr(), g.data_ptr(), B,
                                                       n1, n2, K, g1.data_ptr(), g2.data_ptr(), _stream()),
                       "knn_points_grad")
        return g1, g2, None


def knn_points(p1, p2, K, return_nn=False):
    """pytorch3d.ops.knn.knn_points on dense equal-length clouds:
    (B,n1,3), (B,n2,3) -> (dists (B,n1,K) f32 squared ascending, idx (B,n1,K) i64, nn (B,n1,K,3) | None).
    With grad enabled and an input that requires it, `dists` (and `nn`, as a gather of p2) are differentiable."""
    _req(p1, "p1", torch.float32)
    _req(p2, "p2", torch.float32)
    _same_device(p1, p2)
    if p1.shape[2] != 3 or p2.shape[2] != 3:
        raise RuntimeError("knn_points: only D=3 is built")
    if torch.is_grad_enabled() and (p1.requires_grad or p2.requires_grad):
        dists, idx = _KnnDists.apply(p1, p2, int(K))
        nn = None
        if return_nn:
            B, n1, _ = p1.shape
            safe = idx.clamp(min=0).reshape(B, n1 * int(K), 1).expand(-1, -1, 3)
            nn = p2.gather(1, safe).reshape(B, n1, int(K), 3) * (idx >= 0).unsqueeze(-1)
        return dists, idx, nn
    return _knn_forward(p1, p2, K, return_nn)
