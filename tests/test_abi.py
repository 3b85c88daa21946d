"""The C-ABI library loads and exports every symbol include/pdr_hip.h declares (no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pdr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pdr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    lib = ctypes.CDLL(os.path.join(ROOT, "point_diffusion_refinement_amd", "libpdr_hip.so"))
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libpdr_hip.so does not export %s" % s


def test_python_binding_covers_header():
    from point_diffusion_refinement_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    lib = _lib.load()
    assert lib.pdr_version() == 100


def test_opt_n_threads_matches_reference_table():
    # T(x)=min(2^floor(log2 x),512): 2048->512, 3072->512, 1024->512, 256->256, 64->64, 16->16
    from point_diffusion_refinement_amd import _lib
    from oracle import pdr_oracle as O
    lib = _lib.load()
    for n, t in [(2048, 512), (3072, 512), (1024, 512), (256, 256), (64, 64), (16, 16), (1, 1), (3, 2), (100, 64)]:
        assert lib.pdr_opt_n_threads(n) == t == O.opt_n_threads(n)


def test_workspace_queries_need_no_gpu():
    from point_diffusion_refinement_amd import _lib
    lib = _lib.load()
    assert lib.pdr_fps_workspace_bytes(32, 2048) == 0
    assert lib.pdr_fps_workspace_bytes(2, 20000) == 2 * 20000 * 4
    assert lib.pdr_emd_workspace_bytes(2, 2048, 2048) == 2 * (2048 * 12 + 2048 * 11) * 4
    assert lib.pdr_matchcost_workspace_bytes(2, 2048, 2048) == 2 * 8 * 4


def test_cpu_tensors_are_rejected_not_emulated():
    """No CPU fallback in the product path (reference: AT_ASSERT 'CPU not supported')."""
    import torch
    from point_diffusion_refinement_amd.pointnet2_ops import _ext
    x = torch.rand(1, 16, 3)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        _ext.furthest_point_sampling(x, 4)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        _ext.ball_query(x, x, 0.1, 4)
