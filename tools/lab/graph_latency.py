"""hipGraph dependency latencies on this box, untraced: chains of 1-thread pdr_mark_time kernels (100 MHz wall clock)
captured into a graph and replayed.  (a) N dependent kernels on ONE stream; (b) ping-pong between two streams through
events; (c) the fork / join shape the fused blocks use.  Prints the median spacing of consecutive stamps.
    python -m tools.lab.graph_latency"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_diffusion_refinement_amd import _lib  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    N = 64
    buf = torch.zeros(4 * N, dtype=torch.int64, device=dev)
    a, b = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def mark(i, st):
        _lib.check(lib.pdr_mark_time(buf.data_ptr() + 8 * i, st.cuda_stream), "mark")

    def chain():
        for i in range(N):
            mark(i, torch.cuda.current_stream())

    def pingpong():
        cur = torch.cuda.current_stream()
        a.wait_stream(cur)
        for i in range(N):
            st, other = (a, b) if i % 2 == 0 else (b, a)
            mark(i, st)
            other.wait_stream(st)
        cur.wait_stream(a)
        cur.wait_stream(b)

    def forkjoin():
        cur = torch.cuda.current_stream()
        i = 0
        for _ in range(N // 8):
            mark(i, cur); i += 1                      # before fork
            a.wait_stream(cur)
            for _k in range(3):
                mark(i, a); i += 1                    # aux half
            for _k in range(3):
                mark(i, cur); i += 1                  # main half
            cur.wait_stream(a)
            mark(i, cur); i += 1                      # after join

    out = {}
    for name, fn in (("chain", chain), ("pingpong", pingpong), ("forkjoin", forkjoin)):
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        rows = []
        for _ in range(30):
            g.replay()
            torch.cuda.synchronize()
            rows.append(buf[:N].cpu().numpy().astype("int64"))
        t = np.stack(rows) * 0.01
        if name == "forkjoin":
            t = t.reshape(30, N // 8, 8)
            rel = t - t[:, :, :1]
            med = np.median(rel.reshape(-1, 8), axis=0)
            print("forkjoin: us after the pre-fork kernel: aux %s | main %s | after join %.1f" % (
                np.round(med[1:4], 1), np.round(med[4:7], 1), med[7]))
            out[name] = med.tolist()
        else:
            d = np.diff(t, axis=1)
            print("%s: median spacing %.2f us (p10 %.2f, p90 %.2f)" % (name, np.median(d), np.quantile(d, 0.1),
                                                                       np.quantile(d, 0.9)))
            out[name] = float(np.median(d))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print("   replay of %d nodes: %.1f us" % (N, e0.elapsed_time(e1) * 1000 / 20))
    if len(sys.argv) > 1:
        import json
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
