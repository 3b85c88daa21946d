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
    assert lib.pdr_version() == 200


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


def test_dedup_entry_points_validate_their_arguments_without_a_gpu():
    """pdr_dedup_plan / pdr_dedup_sort / pdr_weighted_moments / pdr_patch_rows / pdr_gather_add_tiles: argument errors
    are return codes decided on the host (nothing is launched); an empty batch is a no-op."""
    from point_diffusion_refinement_amd import _lib
    lib = _lib.load()
    p = 0x1000                                      # (never dereferenced: every call below returns before a launch)
    EINVAL = _lib.PDR_EINVAL
    assert lib.pdr_dedup_plan(p, p, 2, 64, 12, p, p, p, p, p, None) == EINVAL          # K not in {8, 16, 32}
    assert lib.pdr_dedup_plan(p, p, 2, 3, 32, p, p, p, p, p, None) == EINVAL           # m K not a multiple of 128
    assert lib.pdr_dedup_plan(None, p, 2, 64, 32, p, p, p, p, p, None) == EINVAL
    assert lib.pdr_dedup_plan(p, p, 0, 64, 32, p, p, p, p, p, None) == _lib.PDR_OK
    assert lib.pdr_dedup_sort(p, 2, 0, p, p, None, None) == EINVAL
    assert lib.pdr_dedup_sort(p, 0, 64, p, p, None, None) == _lib.PDR_OK
    assert lib.pdr_weighted_moments(p, 64, 2, 256, 64, 0, p, p, 65, 64, p, None) == EINVAL   # ptpb < tpb_full + 2
    assert lib.pdr_weighted_moments(p, 32, 2, 256, 64, 0, p, p, 66, 64, p, None) == EINVAL   # ldy < C
    assert lib.pdr_weighted_moments(p, 64, 0, 256, 64, 0, p, p, 66, 64, p, None) == _lib.PDR_OK
    assert lib.pdr_patch_rows(p, 16, None, None, 1, p, 2, 64, 32, p, 32, None, None) == EINVAL   # ldv < D
    assert lib.pdr_patch_rows(p, 32, None, None, 1, p, 0, 64, 32, p, 32, None, None) == _lib.PDR_OK
    ga = (p, 64, 100, p, None, 64, p, None, None, None, None, None, 2, 256, 32, 64, None, 64, p, 0, 0, -1)
    assert lib.pdr_gather_add_tiles(*ga, None, 4, None) == EINVAL                       # no tile flags
    assert lib.pdr_gather_add_tiles(*ga, p, 1, None) == EINVAL                          # partial_tpb < tiles per cloud


def test_round5_entry_points_validate_their_arguments_without_a_gpu():
    """pdr_dedup_prepare / pdr_dedup_probe / pdr_gather_add_tiles_twin / pdr_embed_select / pdr_gn_fold's subset arguments
    / pdr_reverse_step's probe pair: argument errors are decided on the host; and a tile list without its
    length, a list for a tile height other than 128 and a partial stride smaller than the tiles of a batch element are
    rejected by every layer entry point, pdr_fused_layer_f16x3 included."""
    import ctypes as C
    from point_diffusion_refinement_amd import _lib
    lib = _lib.load()
    p, EINVAL, EUNSUP, OK = 0x1000, _lib.PDR_EINVAL, _lib.PDR_EUNSUPPORTED, _lib.PDR_OK
    prep = lambda B, m, K, idx=p: lib.pdr_dedup_prepare(idx, p, p, B, m, K, p, p, p, p, p, p, p, p, p, p, p, p, None, None)
    assert prep(2, 64, 12) == EINVAL and prep(2, 3, 32) == EINVAL and prep(2, 64, 32, None) == EINVAL
    assert prep(2, 64, 32, 0x1004) == EINVAL                      # index rows move as 16-byte pieces
    assert prep(2000, 64, 32) == EUNSUP and prep(0, 64, 32) == OK
    assert lib.pdr_dedup_probe(p, 2, 64, 12, p, None) == EINVAL and lib.pdr_dedup_probe(p, 2, 64, 32, None, None) == EINVAL
    assert lib.pdr_dedup_probe(p, 0, 64, 32, p, None) == OK
    tw = lambda **k: lib.pdr_gather_add_tiles_twin(
        p, 64, 100, p, None, 64, p, None, 2, 256, 32, 64, None, 64, k.get("partial", p), 0, 0, -1, k.get("tv", p),
        k.get("ptpb", 3), k.get("idx0", p), k.get("Yd", p), k.get("ldyd", 64), k.get("wrow0", p), 32.0, None)
    assert tw(tv=None) == EINVAL and tw(idx0=None) == EINVAL and tw(Yd=None) == EINVAL and tw(wrow0=None) == EINVAL
    assert tw(ptpb=2) == EINVAL and tw(ldyd=62) == EINVAL and tw(partial=None) == EINVAL
    assert lib.pdr_embed_select(p, 64, 10, p, 4, 64, p, 32, None) == EINVAL        # ldo < W
    assert lib.pdr_embed_select(p, 64, 10, p, 4, 62, p, 64, None) == EUNSUP        # W not a multiple of 4
    assert lib.pdr_embed_select(p, 64, 10, p, 0, 64, p, 64, None) == OK
    fold = lambda nv0, tm0, nv1=None, tm1=0, part1=None: lib.pdr_gn_fold(
        p, 64, 10, 64, 1.0, part1, 64 if part1 else 0, 10 if part1 else 0, 64 if part1 else 0, 1.0, 2, 64 + (64 if part1 else 0),
        32, 100.0, 1e-5, p, p, p, p, nv0, tm0, nv1, tm1, None)
    assert fold(p, 0) == EINVAL and fold(p, 11) == EINVAL and fold(None, 0, p, 4) == EINVAL   # second source absent
    assert fold(p, 8, p, 11, p) == EINVAL
    assert lib.pdr_reverse_step(p, p, 3, None, p, p, p, p, None, None, None, p, 10, 0, p, None, None) == EINVAL
    assert lib.pdr_reverse_step(p, p, 3, None, p, p, p, p, None, None, None, p, 0, 0, p, p, None) == OK
    # layer entry points: tile subsets
    li = _lib.LayerIn()
    li.n_seg = 1
    li.seg[0].ptr, li.seg[0].C, li.seg[0].ld, li.seg[0].row_div = 0x1000, 64, 64, 1
    li.rows_per_batch = 1024
    args = (C.byref(li), 2048, 64, 0x2000, 4, None, 128, 0x3000, 128, None, 128, None)
    li.tile_list, li.n_tiles = 0x4000, None
    assert lib.pdr_fused_layer_f16x3(*args) == EUNSUP                          # a list without its length
    li.rows_per_batch, li.n_tiles = 64, 0x5000
    assert lib.pdr_fused_layer_f16x3(*args) == EUNSUP                          # 64-row tiles: not a list of 128-row tiles
    li.rows_per_batch, li.tile_list, li.n_tiles, li.partial_tpb = 1024, None, None, 7
    assert lib.pdr_fused_layer_f16x3(*args) == EINVAL                          # 8 tiles per batch element, stride 7
    plain = (C.byref(li), 2048, 64, 0x2000, 128, None, 128, 0x3000, 128, None, 128, None)
    assert lib.pdr_fused_layer(*plain) == EINVAL


def test_options_replace_the_environment_knobs():
    """ABI 0.2.0: the library never reads the environment; its kernel-selection options are set and
    read back through pdr_set_option / pdr_get_option, enumerated by pdr_option_name, validated by name and range."""
    import subprocess
    from point_diffusion_refinement_amd import _lib
    lib = _lib.load()
    names = _lib.option_names()
    assert names == ["fused_ws", "narrow_kc32", "fps_wave", "fps_lean", "knn_wave", "gn_fold_small", "ws_narrow3",
                     "ws_xcd_order", "deep_chunks", "deep_ks", "deep_jobs32", "deep_jobs64"]
    assert lib.pdr_option_name(len(names)) is None and lib.pdr_option_name(-1) is None
    defaults = {n: _lib.get_option(n) for n in names}
    assert all(v == 1 for n, v in defaults.items() if not n.startswith("deep_jobs")) and defaults["deep_jobs32"] == 256
    assert lib.pdr_set_option(b"fps_wave", 3) == _lib.PDR_EINVAL and lib.pdr_set_option(b"fps_wave", -1) == _lib.PDR_EINVAL
    assert lib.pdr_set_option(b"no_such_option", 1) == _lib.PDR_EINVAL and lib.pdr_set_option(None, 1) == _lib.PDR_EINVAL
    assert lib.pdr_get_option(b"fps_wave", None) == _lib.PDR_EINVAL
    _lib.set_option("fps_wave", 2)
    _lib.set_option("ws_xcd_order", 0)
    try:
        assert _lib.get_option("fps_wave") == 2 and _lib.get_option("ws_xcd_order") == 0
        # tile geometry follows the option at once (no "read once per process"): narrow outputs
        assert lib.pdr_fused_layer_tile_rows(65536, 32) == 128
        _lib.set_option("narrow_kc32", 0)
        assert lib.pdr_fused_layer_tile_rows(65536, 32) == 256
    finally:
        for n in names:
            _lib.set_option(n, defaults[n])
    # no getenv in the shipped library
    out = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in out, "libpdr_hip.so imports getenv"


def test_pooled_launch_validates_its_patch_arguments_without_a_gpu():
    """pdr_layer_in_t.patch_values makes the pooled launch read 16-byte pieces of patch_values / patch_w and
    write 16-byte pieces of `out`: a missing weight array, a stride below D, widths / strides that are not multiples of
    four floats and unaligned pointers are argument errors, decided on the host by both pooled entry points."""
    import ctypes as C
    from point_diffusion_refinement_amd import _lib
    lib = _lib.load()

    def call(fn, D=64, ldo=64, out=0x7000, **patch):
        li = _lib.LayerIn()
        li.n_seg = 1
        li.seg[0].ptr, li.seg[0].C, li.seg[0].ld, li.seg[0].row_div = 0x1000, 64, 64, 1
        li.rows_per_batch = 4096
        li.patch_values, li.patch_w, li.patch_ld = patch.get("values", 0x8000), patch.get("w", 0x9000), patch.get("ld", 64)
        if fn == "exact":
            return lib.pdr_fused_layer_pool(C.byref(li), 0, 64, 0x2000, 64, None, D, 0x3000, 64, None, None, 1, None, 32,
                                            out, ldo, None)
        return lib.pdr_fused_layer_pool_f16x3(C.byref(li), 0, 64, 0x2000, 2, None, D, 0x3000, 64, None, None, 1, None, 32,
                                              out, ldo, None)
    for fn in ("exact", "split"):
        assert call(fn, w=None) == _lib.PDR_EINVAL
        assert call(fn, ld=60) == _lib.PDR_EINVAL and call(fn, ld=66) == _lib.PDR_EINVAL
        assert call(fn, values=0x8004) == _lib.PDR_EINVAL and call(fn, out=0x7008) == _lib.PDR_EINVAL
        assert call(fn, D=62, ldo=64) == _lib.PDR_EINVAL and call(fn, ldo=66) == _lib.PDR_EINVAL


def test_plan_reports_right_sized_tiny_layers():
    """pdr_fused_layer_plan is host logic (no launch): out[7] = 128 exactly for the tiny layers of DESIGN 4.9 -- more than
    64 input channels and few enough row tiles that every workgroup of the launch is resident at once (32-row tiles:
    <= 256 jobs of 32 x 128; 64-row tiles, plain sources: <= 512 jobs of 64 x 64; 128-row tiles, plain sources: <= 128
    jobs) -- and out[0] (wave-specialised kernel) is 0 for the 64- / 128-row ones."""
    import ctypes as C
    from point_diffusion_refinement_amd import _lib
    lib = _lib.load()

    def plan(B, rpb, Cin, Cout, gathered=False):
        li = _lib.LayerIn()
        li.n_seg = 1
        ld = (Cin + 3) // 4 * 4
        li.seg[0].ptr, li.seg[0].C, li.seg[0].ld, li.seg[0].row_div = 0x10000, Cin, ld, 1
        li.rows_per_batch = rpb
        if gathered:
            li.seg[0].gV, li.seg[0].gV0, li.seg[0].g_ldv, li.seg[0].g_nsrc, li.seg[0].g_zrow = 0x20000, 0x20000 + 4 * ld, 2 * ld, 64, B * 64
            li.gidx, li.gcnt, li.gK = 0x30000, 0x40000, 16
        out = (C.c_int * 8)()
        ldw = (Cout + 3) // 4 * 4
        assert lib.pdr_fused_layer_plan(C.byref(li), B * rpb, Cin, 0x50000, ldw, Cout, 0x60000, ldw, out) == 0
        return list(out)

    for B, rpb, Cin, Cout, deep in [(32, 16, 512, 512, True), (32, 16, 643, 1163, False), (32, 16, 64, 512, False),
                                    (32, 64, 256, 256, True), (32, 64, 323, 1097, False), (32, 256, 128, 128, True),
                                    (32, 512, 128, 128, True), (32, 512, 256, 256, False), (32, 2048, 128, 128, False)]:
        p = plan(B, rpb, Cin, Cout)
        assert (p[7] == 128) == deep, (B, rpb, Cin, Cout, p)
        if deep and rpb >= 64:
            assert p[0] == 0, p                                   # not the wave-specialised kernel
        if not deep and rpb >= 64:
            assert p[0] == 1, p
    assert plan(32, 64, 256, 256, gathered=True)[7] == 0           # gathered sources: the wave-specialised tiles
    assert plan(32, 256, 128, 128, gathered=True)[7] == 0


def test_round6_entry_points_validate_their_arguments_without_a_gpu():
    """pdr_point_chain / pdr_point_chain_plan / pdr_fused_layer_pair: argument errors and unsupported shapes are decided
    on the host (no launch); the plan reports the cluster size and the scratch a chain needs."""
    import ctypes as C
    from point_diffusion_refinement_amd import _lib
    lib = _lib.load()
    EINVAL, EUNSUP, OK = _lib.PDR_EINVAL, _lib.PDR_EUNSUPPORTED, _lib.PDR_OK

    def chain(n_layers=2, seg=(64, 64), widths=((128, 256), (128, 128)), residual=1, groups=32, cn=None):
        ch = _lib.PointChain()
        ch.n_layers, ch.n_seg, ch.residual = n_layers, 1, residual
        ch.seg[0].ptr, ch.seg[0].C, ch.seg[0].ld = 0x1000, seg[0], seg[1]
        cin = seg[0]
        for i, (main, cout) in enumerate(widths[:n_layers]):
            L = ch.layer[i]
            L.Wt, L.bias, L.ldw, L.Cin, L.Cout, L.main_cols = 0x2000, 0x3000, cout, cin, cout, main
            L.gamma, L.beta, L.groups, L.Cn, L.eps, L.relu_post = 0x4000, 0x5000, groups, cn or main, 1e-5, 1
            cin = main
        return ch
    plan = (C.c_long * 4)()
    ch = chain()
    assert lib.pdr_point_chain_plan(C.byref(ch), 32, 64, plan) == OK
    assert list(plan) == [8, 32 * 64 * 128, 65, 256]                # G, floats of scratch, ints of sync, workgroups
    assert lib.pdr_point_chain_plan(C.byref(ch), 64, 64, plan) == OK and plan[0] == 4     # B G <= 256
    assert lib.pdr_point_chain_plan(C.byref(ch), 32, 512, plan) == EUNSUP                  # rows per cloud
    assert lib.pdr_point_chain_plan(C.byref(ch), 32, 48, plan) == EUNSUP
    assert lib.pdr_point_chain_plan(C.byref(chain(widths=((120, 248), (128, 128)), cn=96)), 32, 64, plan) == EUNSUP   # 120 columns
    assert lib.pdr_point_chain_plan(C.byref(chain(widths=((128, 256), (64, 64)))), 32, 64, plan) == EINVAL   # residual width
    assert lib.pdr_point_chain_plan(C.byref(chain(seg=(64, 62))), 32, 64, plan) == EINVAL                    # ld < C
    assert lib.pdr_point_chain_plan(C.byref(chain(n_layers=0)), 32, 64, plan) == EINVAL
    bad = chain()
    bad.layer[1].Cin = 100
    assert lib.pdr_point_chain_plan(C.byref(bad), 32, 64, plan) == EINVAL                                     # chain of widths
    assert lib.pdr_point_chain(C.byref(ch), 32, 64, None) == EINVAL                                           # no out / scratch / sync
    assert lib.pdr_point_chain(None, 32, 64, None) == EINVAL
    # paired launch: both problems validated like pdr_fused_layer; a first problem without a tile list is not a pair
    li, li2 = _lib.LayerIn(), _lib.LayerIn()
    for x, rpb in ((li, 2048), (li2, 64)):
        x.n_seg = 1
        x.seg[0].ptr, x.seg[0].C, x.seg[0].ld, x.seg[0].row_div = 0x1000, 64, 64, 1
        x.rows_per_batch = rpb
    args = lambda a, b, P2=128: (C.byref(a), 4096, C.byref(b), P2, 64, 0x2000, 64, None, 64, 0x3000, 64, 0x4000, 64,
                                 None, None, 64, None)
    assert lib.pdr_fused_layer_pair(*args(li, li2)) == EUNSUP                  # no tile list
    assert lib.pdr_fused_layer_pair(*args(li, li2, P2=100)) == EINVAL          # P2 not a multiple of its rows per cloud
    li.tile_list, li.n_tiles = 0x5000, 0x6000
    li2.tile_list, li2.n_tiles = 0x5000, 0x6000
    assert lib.pdr_fused_layer_pair(*args(li, li2)) == EUNSUP                  # a listed second problem
    assert lib.pdr_fused_layer_pair(None, 4096, C.byref(li2), 128, 64, 0x2000, 64, None, 64, 0x3000, 64, 0x4000, 64, None,
                                    None, 64, None) == EINVAL
