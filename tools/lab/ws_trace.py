"""Per-chunk timeline of one consumer and one producer wave of the wave-specialised layer kernel
(lab build with -DPDR_LAB_TRACE):  python -m tools.lab.ws_trace [shape index]"""
import ctypes
import sys

import numpy as np
import torch

from point_diffusion_refinement_amd import _lib
from tools import fused_layer_bench as FB


def main():
    idx = int(sys.argv[1]) if len(sys.argv) > 1 else 13
    _lib.LIB_PATH = __import__("os").environ.get("PDR_LAB_LIB", "point_diffusion_refinement_amd/libpdr_lab.so")
    lib = _lib.load()
    extra = sys.argv[2:]                                  # e.g. --gath 32
    sys.argv = [sys.argv[0], "--only", str(idx), "--reps", "1", "--lib", _lib.LIB_PATH] + extra
    FB.main()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    buf = np.zeros((3, 4096), dtype=np.uint64)
    assert raw.pdr_lab_trace_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    if hasattr(raw, "pdr_lab_wg_read"):
        wg = np.zeros((2048, 3), dtype=np.uint64)
        assert raw.pdr_lab_wg_read(wg.ctypes.data_as(ctypes.c_void_p)) == 0
        wg = wg[wg[:, 1] > 0].astype(np.int64)
    if hasattr(raw, "pdr_lab_wg_read") and len(wg):
        t00 = wg[:, 0].min()
        st, en, dur = (wg[:, 0] - t00) / 100.0, (wg[:, 1] - t00) / 100.0, (wg[:, 1] - wg[:, 0]) / 100.0
        xcc, tiles = wg[:, 2] & 15, wg[:, 2] >> 8
        q = lambda a: "min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (
            a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
        print("workgroups of the last launch (us on the 100 MHz device clock): %d, tiles each %d..%d" %
              (len(wg), tiles.min(), tiles.max()))
        print("  start    ", q(st))
        print("  end      ", q(en))
        print("  duration ", q(dur))
        for x in range(8):
            m = xcc == x
            if m.any():
                print("  XCC %d: %3d workgroups, start median %.1f, duration median %.1f max %.1f, last end %.1f" %
                      (x, m.sum(), np.median(st[m]), np.median(dur[m]), dur[m].max(), en[m].max()))
    c = buf[0].reshape(-1, 4).astype(np.int64)
    p = buf[1].reshape(-1, 4).astype(np.int64)
    n = int((c[:, 0] > 0).sum())
    t0 = min(c[0, 0], p[0, 0])
    print("chunk | consumer: wait  compute  epi | producer: loadwait  commit  fetch-issue | c.period | p.arrive-c.arrive")
    e = buf[2].reshape(-1, 8).astype(np.int64)
    rpb, Cin, Cout = FB.SHAPES[idx]
    kc = 16 if (lib.pdr_fused_layer_variant(rpb, Cout) in (0, 1)) else 32
    nch = (Cin + kc - 1) // kc
    print("epilogue checkpoints per tile (ticks after the last MFMA issue; slot 0: end of column tile 0, 1: Tt writes of "
          "half 0, 2: statistics of half 0, 3 / 4: stores of half 0 / 1, 6: before the ticket; c3 = end of epilogue)")
    for t in range(min(6, n // nch)):
        base = c[t * nch + nch - 1, 2]
        print("  tile %d:" % t, {k: int(x - base) for k, x in enumerate(e[t]) if x > 0}, "end", int(c[t * nch + nch - 1, 3] - base))
    for g in range(min(n, 70)):
        per = c[g + 1, 0] - c[g, 0] if g + 1 < n else 0
        print("%4d | %6d %6d %6d | %6d %6d %6d | %6d | %d" % (
            g, c[g, 1] - c[g, 0], c[g, 2] - c[g, 1], c[g, 3] - c[g, 2],
            p[g, 3] - p[g, 0], p[g, 1] - p[g, 3], p[g, 2] - p[g, 1], per, p[g, 0] - c[g, 0]))


def epilogue_detail():
    """second table: per tile, time from the end of the MFMA loop to each epilogue checkpoint"""


if __name__ == "__main__":
    main()

