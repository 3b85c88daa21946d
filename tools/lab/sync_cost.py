"""What a dependent launch costs inside a replayed hipGraph: N tiny kernels as (a) one chain on one stream, (b) a chain
that hops between two streams at every link (an event record + wait per link), (c) two independent chains on two
streams.   python -m tools.lab.sync_cost
Round 6, MI355X: 1.65 / 1.62 us per kernel for (a) / (b) -- a cross-stream dependence costs what a same-stream one costs --,
3.28 us per kernel for (c): two independent chains of one-thread kernels do not overlap in a replayed graph (nor do two
graphs in flight: tools/lab/two_batches.py).  A four-chain version of (c) segfaulted at capture (ROCm 7.2).  The real
step's two block-half streams DO pay: bench.py --single-stream 6.70 vs 5.61 ms."""
import torch

from point_diffusion_refinement_amd import _lib


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    buf = torch.zeros(4096, dtype=torch.int64, device=dev)
    N = 200

    def tiny(i):
        _lib.check(lib.pdr_mark_time(buf.data_ptr() + 8 * (i % 4096), torch.cuda.current_stream().cuda_stream), "mark")

    def chain_one():
        for i in range(N):
            tiny(i)

    s2 = torch.cuda.Stream()

    def chain_hop():
        main = torch.cuda.current_stream()
        for i in range(N):
            st = main if i % 2 == 0 else s2
            other = s2 if i % 2 == 0 else main
            with torch.cuda.stream(st):
                tiny(i)
                ev = torch.cuda.Event()
                ev.record(st)
            other.wait_event(ev)
        main.wait_stream(s2)

    def two_chains():
        main = torch.cuda.current_stream()
        s2.wait_stream(main)
        with torch.cuda.stream(s2):
            for i in range(N // 2):
                tiny(i)
        for i in range(N // 2):
            tiny(i + 2000)
        main.wait_stream(s2)

    for name, fn in (("one stream", chain_one), ("hop every link", chain_hop), ("two independent chains", two_chains)):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(10):
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print("%-24s %4d one-thread kernels: %7.1f us per replay = %.2f us per kernel" %
              (name, N, sorted(ts)[5], sorted(ts)[5] / N))


if __name__ == "__main__":
    main()
