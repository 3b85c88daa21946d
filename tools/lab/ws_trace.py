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
    sys.argv = [sys.argv[0], "--only", str(idx), "--reps", "1", "--lib", _lib.LIB_PATH]
    FB.main()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    buf = np.zeros((3, 4096), dtype=np.uint64)
    assert raw.pdr_lab_trace_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    c = buf[0].reshape(-1, 4).astype(np.int64)
    p = buf[1].reshape(-1, 4).astype(np.int64)
    n = int((c[:, 0] > 0).sum())
    t0 = min(c[0, 0], p[0, 0])
    print("chunk | consumer: wait  compute  epi | producer: loadwait  commit  fetch-issue | c.period | p.arrive-c.arrive")
    e = buf[2].reshape(-1, 8).astype(np.int64)
    nch = 4
    print("epilogue checkpoints per tile (ticks after the last MFMA issue): after column tile 0, 1, ..., before ticket")
    for t in range(min(6, n // nch)):
        base = c[t * nch + nch - 1, 2]
        print("  tile %d:" % t, [int(x - base) for x in e[t] if x > 0])
    for g in range(min(n, 70)):
        per = c[g + 1, 0] - c[g, 0] if g + 1 < n else 0
        print("%4d | %6d %6d %6d | %6d %6d %6d | %6d | %d" % (
            g, c[g, 1] - c[g, 0], c[g, 2] - c[g, 1], c[g, 3] - c[g, 2],
            p[g, 3] - p[g, 0], p[g, 1] - p[g, 3], p[g, 2] - p[g, 1], per, p[g, 0] - c[g, 0]))


def epilogue_detail():
    """second table: per tile, time from the end of the MFMA loop to each epilogue checkpoint"""


if __name__ == "__main__":
    main()

