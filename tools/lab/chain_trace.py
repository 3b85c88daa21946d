"""Phase timeline of one workgroup of pdr_point_chain (lab build -DPDR_LAB_TRACE -> libpdr_lab.so):
    bash tools/lab/build_lab.sh -DPDR_LAB_TRACE && python -m tools.lab.chain_trace"""
import ctypes
import os

import numpy as np
import torch

from point_diffusion_refinement_amd import _lib

_lib.LIB_PATH = os.environ.get("PDR_LAB_LIB", "point_diffusion_refinement_amd/libpdr_lab.so")


def main():
    from tests.test_fused_gpu import _chain_case
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    dev = torch.device("cuda:0")
    for B, n, seg_widths, widths in ((32, 64, (256, 256, 3), (256, 256)), (32, 256, (128, 256, 3), (256, 256))):
        segs, layers, _ = _chain_case(5, B, n, seg_widths, widths, True, False, dev)
        keep = [s.to(dev) for s in segs]
        ch = _lib.PointChain()
        ch.n_layers, ch.n_seg, ch.residual = len(layers), len(segs), 1
        for i, (t, c) in enumerate(zip(keep, seg_widths)):
            ch.seg[i].ptr, ch.seg[i].C, ch.seg[i].ld = t.data_ptr(), c, t.shape[1]
        for i, Ld in enumerate(layers):
            L = ch.layer[i]
            dv = {k: (v.to(dev).contiguous() if torch.is_tensor(v) else v) for k, v in Ld.items()}
            keep.append(dv)
            L.Wt, L.bias, L.ldw, L.Cin, L.Cout, L.main_cols = dv["W"].data_ptr(), dv["bias"].data_ptr(), dv["cout"], \
                dv["W"].shape[0], dv["cout"], dv["c"]
            L.gamma, L.beta, L.groups, L.Cn, L.eps, L.relu_post = dv["gamma"].data_ptr(), dv["beta"].data_ptr(), 32, dv["cn"], 1e-5, 1
            L.add, L.add_ld = dv["add"].data_ptr(), dv["add"].shape[1]
        plan = (ctypes.c_long * 4)()
        assert lib.pdr_point_chain_plan(ctypes.byref(ch), B, n, plan) == 0
        out = torch.empty((B * n, widths[-1]), device=dev)
        scratch = torch.empty(int(plan[1]), device=dev)
        sync = torch.zeros(int(plan[2]), dtype=torch.int32, device=dev)
        ch.out, ch.ldo, ch.scratch, ch.sync = out.data_ptr(), out.shape[1], scratch.data_ptr(), sync.data_ptr()
        for _ in range(5):
            _lib.check(lib.pdr_point_chain(ctypes.byref(ch), B, n, None), "chain")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            _lib.check(lib.pdr_point_chain(ctypes.byref(ch), B, n, None), "chain")
        e1.record()
        torch.cuda.synchronize()
        buf = np.zeros(64, dtype=np.uint64)
        assert raw.pdr_lab_chain_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        t = buf.astype(np.int64)
        names = ["first chunk staged", "chunk loop done", "stats + fold done", "(store begin)", "stores issued", "drained + barrier",
                 "counter reached", "acquired"]
        print("n = %d, G = %d: %.1f us per launch (20 eager launches back to back)" % (n, plan[0], e0.elapsed_time(e1) / 20 * 1e3))
        prev = t[0]
        for l in range(len(layers)):
            for k in range(8):
                v = t[1 + 8 * l + k - 0] if (1 + 8 * l + k) < 64 else 0
                slot = 1 + 8 * l + k
                if t[slot] > 0 and t[slot] >= prev:
                    print("  layer %d %-20s +%7d cycles (%.2f us)  at %.2f us" %
                          (l, names[k], t[slot] - prev, (t[slot] - prev) / 2400.0, (t[slot] - t[0]) / 2400.0))
                    prev = t[slot]
        print("  layer 0, chunks 0-7 (cycles): fetch issue | multiply | commit (wait + LDS writes) | barrier")
        for c in range(8):
            a = t[32 + 4 * c:36 + 4 * c]
            nxt = t[36 + 4 * c] if c < 7 else 0
            if a.min() > 0:
                print("   chunk %d: %6d | %6d | %6d | %6d" % (c, a[1] - a[0], a[2] - a[1], a[3] - a[2],
                                                            (nxt - a[3]) if nxt > 0 else -1))


if __name__ == "__main__":
    main()
